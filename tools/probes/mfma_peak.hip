// What does v_mfma_f32_32x32x16_f16 sustain on this chip, with nothing else in the loop?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o build_variants/mfma_peak ; run on the GPU box
// Variants: CHAINS independent accumulators per wave (1 = every MFMA depends on the previous one),
// WAVES per block (1 block per CU): 4 / 8 / 12 = 1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ void mfma_loop(float *out, int iters, unsigned long long *clk) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(0.001f * (threadIdx.x + i));   // non-trivial operands (data-dependent power)
    b[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i));
  }
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

template <int CHAINS>
void run(int waves, int iters) {
  float *out;
  unsigned long long *clk, h[2];
  hipMalloc(&out, 256 * 1024 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(256), dim3(64 * waves), 0, 0, out, iters, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = 256.0 * waves * iters * 8 * CHAINS;
  const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  printf("chains %d  waves/CU %2d : %8.3f ms  %7.1f TFLOP/s (%.1f %% of 2500)  shader clock %.2f GHz  cycles per MFMA per SIMD %.1f\n",
         CHAINS, waves, ms, tf, 100 * tf / 2500, ghz, ms * 1e-3 * ghz * 1e9 / (mfmas / 1024.0));
}

int main() {
  const int iters = 20000;
  for (int w : {4, 8, 12}) {
    run<1>(w, iters);
    run<2>(w, iters);
    run<4>(w, iters);
    run<8>(w, iters);
  }
  return 0;
}
