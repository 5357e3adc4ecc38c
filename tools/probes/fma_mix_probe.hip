// probe: v_cvt_pk_f16_f32 + v_fma_mix_f32 half selection (attention P split)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float *in, float *out) {
  const float p0 = in[2 * threadIdx.x], p1 = in[2 * threadIdx.x + 1];
  unsigned hh, ll;
  float d0, d1;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(p0), "v"(p1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hh), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hh), "v"(p1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ll) : "v"(d0), "v"(d1));
  const _Float16 h0 = (_Float16)p0, h1 = (_Float16)p1;
  out[8 * threadIdx.x + 0] = d0;
  out[8 * threadIdx.x + 1] = p0 - (float)h0;
  out[8 * threadIdx.x + 2] = d1;
  out[8 * threadIdx.x + 3] = p1 - (float)h1;
  out[8 * threadIdx.x + 4] = __builtin_bit_cast(float, hh);
  out[8 * threadIdx.x + 5] = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16));
  out[8 * threadIdx.x + 6] = __builtin_bit_cast(float, ll);
  out[8 * threadIdx.x + 7] = 0.f;
}
int main() {
  float h_in[8] = {0.7231f, 0.0123456f, 1.0f, 3.3e-5f, 0.333333f, 0.99999f, 0.5f, 1e-7f}, h_out[32];
  float *d_in, *d_out;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; ++t)
    printf("p0 %.7g p1 %.7g | d0 %.6e (want %.6e) d1 %.6e (want %.6e) | hh %08x want %08x ll %08x\n", h_in[2 * t], h_in[2 * t + 1],
           h_out[8 * t], h_out[8 * t + 1], h_out[8 * t + 2], h_out[8 * t + 3], *(unsigned *)&h_out[8 * t + 4],
           *(unsigned *)&h_out[8 * t + 5], *(unsigned *)&h_out[8 * t + 6]);
  return 0;
}
