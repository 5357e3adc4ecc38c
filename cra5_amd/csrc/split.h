// Split-f16 element store shared by the producers (see gemm_split_f16.hip for the format):
// row = [chunk 0: 32 hi halves | 32 lo halves][chunk 1: ...]; x = hi + lo, hi = f16(x),
// lo = f16(x - hi).
//
// Range: f16 tops out at 65 504.  The value is SATURATED there before the split (one v_med3 per
// element), so an out-of-range activation degrades to +-65 504 instead of hi = inf, lo = -inf ->
// NaN products.  Builds with
// -DCRA5_RANGE_CHECK (python -m cra5_amd.build --flavour rangecheck) additionally count, per
// producer call site, the elements with |x| >= 65 504 and the non-finite ones
// (cra5_debug_range_counts in the C ABI): the evidence that a checkpoint's activations stay
// inside the exactly-represented range.
#pragma once
#include <hip/hip_runtime.h>

#ifdef CRA5_RANGE_CHECK
// one counter pair per translation unit (no relocatable device code needed); each .hip that stores
// split values instantiates CRA5_RANGE_TU(name), and cra5_debug_range_counts (elementwise.hip) sums them
static __device__ unsigned long long g_cra5_range_counts[2];   // [0] |x| >= 65504, [1] non-finite
__device__ __forceinline__ void cra5_range_probe(float v) {
  const float a = __builtin_fabsf(v);
  if (!(a < 65504.0f)) atomicAdd(&g_cra5_range_counts[(a == a && a != __builtin_inff()) ? 0 : 1], 1ULL);
}
#define CRA5_RANGE_TU(name)                                                                          \
  extern "C" __attribute__((visibility("hidden"))) int cra5_range_counts_##name(unsigned long long *acc, int reset) { \
    unsigned long long h[2] = {0, 0};                                                                \
    int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cra5_range_counts), sizeof(h));                \
    if (rc) return rc;                                                                               \
    acc[0] += h[0];                                                                                  \
    acc[1] += h[1];                                                                                  \
    if (reset) {                                                                                     \
      const unsigned long long z[2] = {0, 0};                                                        \
      rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cra5_range_counts), z, sizeof(z));                    \
    }                                                                                                \
    return rc;                                                                                       \
  }
#else
__device__ __forceinline__ void cra5_range_probe(float) {}
#define CRA5_RANGE_TU(name)
#endif

__device__ __forceinline__ void cra5_split(float v, _Float16 &hi, _Float16 &lo) {
  cra5_range_probe(v);
  const float vc = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
  hi = (_Float16)vc;
  lo = (_Float16)(vc - (float)hi);
}

__device__ __forceinline__ void cra5_store_split(unsigned short *row, int n, float v) {
  _Float16 hi, lo;
  cra5_split(v, hi, lo);
  unsigned short *p = row + (n >> 5) * 64 + (n & 31);
  p[0] = __builtin_bit_cast(unsigned short, hi);
  p[32] = __builtin_bit_cast(unsigned short, lo);
}

// four consecutive columns n..n+3 (n % 4 == 0): two 8-byte stores
__device__ __forceinline__ void cra5_store_split4(unsigned short *row, int n, float a, float b, float c, float d) {
  _Float16 h0, h1, h2, h3, l0, l1, l2, l3;
  cra5_split(a, h0, l0);
  cra5_split(b, h1, l1);
  cra5_split(c, h2, l2);
  cra5_split(d, h3, l3);
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  half4 hv = {h0, h1, h2, h3}, lv = {l0, l1, l2, l3};
  unsigned short *p = row + (n >> 5) * 64 + (n & 31);
  *reinterpret_cast<half4 *>(p) = hv;
  *reinterpret_cast<half4 *>(p + 32) = lv;
}
