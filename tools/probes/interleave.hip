// How many VALU instructions hide behind a v_mfma_f32_32x32x16_f16 when they follow it IN THE SAME WAVE,
// with 1, 2 or 3 such waves per SIMD?   hipcc --offload-arch=gfx950 -O3 tools/probes/interleave.hip -o build_variants/interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ void k(float *out, int iters) {
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  half8 a, b;
  float v[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));
    b[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i));
    v[i] = 0.001f * (threadIdx.x + i);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[(u + q) & 7] = __builtin_fmaf(v[(u + q) & 7], 0.999f, 0.001f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run(int waves) {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(64 * waves), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double mfma_per_simd = (double)iters * 8 * waves / 4;
  printf("  %d VALU per MFMA, %2d waves/CU: %7.3f ms  %6.1f ns per MFMA per SIMD  (%.0f TFLOP/s of MFMA)\n", NV, waves, ms,
         ms * 1e6 / mfma_per_simd, 256.0 * waves * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int w : {4, 8, 12}) {
    run<0>(w); run<2>(w); run<4>(w); run<6>(w); run<8>(w); run<12>(w);
  }
  return 0;
}
