// Do MFMA and VALU streams of DIFFERENT waves on one SIMD overlap?  Waves 0-3 of every block (one per SIMD)
// run an MFMA loop, waves 4-7 a VALU loop (v_fma_f32 / v_exp_f32 mix); measured alone and together.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coissue.hip -o build_variants/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int valu_iters, unsigned long long *clk) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = wall_clock64();
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
      a[i] = (_Float16)(0.001f * (threadIdx.x + i));
      b[i] = (_Float16)(0.002f * (threadIdx.x * 3 + i));
    }
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#ifdef ACC_AGPR   // accumulators pinned to the AGPR half of the register file
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(a), "v"(b));
#else
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
#endif
      }
    }
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 16; ++r) s += acc[c][r];
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);   // 32 independent-ish FMAs
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] - 1.0f);       // 2 transcendentals
    }
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  float *out;
  unsigned long long *clk;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&clk, 256 * 8 * 8);
  static unsigned long long h[256 * 8];
  const int MI = 4000, VI = 4000;   // 64 000 MFMAs per MFMA wave; 136 000 VALU instructions per VALU wave
  const int cfg[3][2] = {{MI, 0}, {0, VI}, {MI, VI}};
  const char *name[3] = {"MFMA waves only", "VALU waves only", "both"};
  for (int c = 0; c < 3; ++c) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cfg[c][0], cfg[c][1], clk);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < 256; ++b) {
      for (int w = 0; w < 4; ++w) tm += h[b * 8 + w];
      for (int w = 4; w < 8; ++w) tv += h[b * 8 + w];
    }
    tm /= 1024 * 100.0;   // us (100 MHz wall clock)
    tv /= 1024 * 100.0;
    printf("%-16s: MFMA waves %8.1f us  VALU waves %8.1f us", name[c], tm, tv);
    if (cfg[c][0]) printf("   -> %.1f TFLOP/s of MFMA", 1024.0 * cfg[c][0] * 16 * 32768.0 / (tm * 1e-6) / 1e12);
    printf("\n");
  }
  return 0;
}
