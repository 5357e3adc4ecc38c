// Split-f16 element store shared by the producers (see gemm_split_f16.hip for the format):
// row = [chunk 0: 32 hi halves | 32 lo halves][chunk 1: ...]; x = hi + lo, hi = f16(x),
// lo = f16(x - hi).
//
// Range: f16 tops out at 65 504.  An element beyond it is NOT clipped (round 4; rounds 1-3 saturated at +-65 504, which
// kept such a frame finite and wrong - silently): `hi = f16(x)` overflows to +-inf, `lo = f16(x - hi)` to -+inf, and
// every product that reads the pair is inf - inf = NaN.  The poison reaches the fp32 output of the consuming GEMM /
// attention row, LayerNorm and the global attention spread it over the frame, and the model's finiteness probe at the
// end of the GPU phase (cra5_amd/vaeformer.py: _range_guard) re-runs that frame on the exact-f32 engines - or raises in
// the hyper-prior path, whose engine is pinned on both sides of the codec.  Out-of-range activations can therefore
// never degrade a frame quietly, and the in-range path carries no range-handling instruction at all (the saturating
// form cost a v_med3 + a v_fma per element in every producer).  In-range values split exactly as before (bit-identical
// streams).  NaN / inf inputs stay non-finite by the same arithmetic.  Builds with
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
  hi = (_Float16)v;                 // |v| >= 65 520: +-inf
  lo = (_Float16)(v - (float)hi);   // then -+inf (NaN for a non-finite v): the pair poisons every product it enters
}

// Two values -> packed (hi, hi) and (lo, lo) f16 pairs.  lo = x - f32(hi) is ONE v_fma_mix_f32 reading the f16 half
// in place (no v_cvt_f32_f16 + v_sub); the conversions on both sides stay compiler-generated (packed
// v_cvt_pk_f16_f32).  `s_nop 0`: a and b may come straight out of a transcendental (v_exp_f32 / v_rcp_f32 in the GELU
// epilogue), and gfx950 wants a wait state between a transcendental and a VALU that reads its result - hipcc inserts
// it for its own instructions, not for inline asm (found the hard way in attention_split_f16.hip).
__device__ __forceinline__ void cra5_split_pair(float a, float b, unsigned &hi2, unsigned &lo2) {
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  cra5_range_probe(a);
  cra5_range_probe(b);
  const float ac = a, bc = b;
  const half2v h2 = {(_Float16)ac, (_Float16)bc};
  hi2 = __builtin_bit_cast(unsigned, h2);
#if defined(__HIP_DEVICE_COMPILE__)
  float d0, d1;
  asm("s_nop 0\n\tv_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(d0), "=&v"(d1) : "v"(hi2), "v"(ac), "v"(bc));
#else
  const float d0 = ac - (float)h2[0], d1 = bc - (float)h2[1];
#endif
  const half2v l2 = {(_Float16)d0, (_Float16)d1};
  lo2 = __builtin_bit_cast(unsigned, l2);
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
  unsigned h01, l01, h23, l23;
  cra5_split_pair(a, b, h01, l01);
  cra5_split_pair(c, d, h23, l23);
  unsigned short *p = row + (n >> 5) * 64 + (n & 31);
  *reinterpret_cast<uint2 *>(p) = make_uint2(h01, h23);
  *reinterpret_cast<uint2 *>(p + 32) = make_uint2(l01, l23);
}

// PLAIN f16 rows (reduced-precision mode, round 5): element n of a row is the half at offset n - the same hi value the
// split layout stores, no lo plane.  Four consecutive columns (n % 4 == 0): one 8-byte store.
__device__ __forceinline__ void cra5_store_plain4(unsigned short *row, int n, float a, float b, float c, float d) {
  unsigned h01, l01, h23, l23;
  cra5_split_pair(a, b, h01, l01);   // (the lo halves are dead code here; the range probe of the rangecheck flavour stays)
  cra5_split_pair(c, d, h23, l23);
  *reinterpret_cast<uint2 *>(row + n) = make_uint2(h01, h23);
}
__device__ __forceinline__ void cra5_store_plain(unsigned short *row, int n, float v) {
  _Float16 hi, lo;
  cra5_split(v, hi, lo);
  row[n] = __builtin_bit_cast(unsigned short, hi);
}
