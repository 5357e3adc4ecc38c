// Split-f16 element store shared by the producers (see gemm_split_f16.hip for the format):
// row = [chunk 0: 32 hi halves | 32 lo halves][chunk 1: ...]; x = hi + lo, hi = f16(x),
// lo = f16(x - hi).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void cra5_store_split(unsigned short *row, int n, float v) {
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  unsigned short *p = row + (n >> 5) * 64 + (n & 31);
  p[0] = __builtin_bit_cast(unsigned short, hi);
  p[32] = __builtin_bit_cast(unsigned short, lo);
}

// four consecutive columns n..n+3 (n % 4 == 0): two 8-byte stores
__device__ __forceinline__ void cra5_store_split4(unsigned short *row, int n, float a, float b, float c, float d) {
  const _Float16 h0 = (_Float16)a, h1 = (_Float16)b, h2 = (_Float16)c, h3 = (_Float16)d;
  const _Float16 l0 = (_Float16)(a - (float)h0), l1 = (_Float16)(b - (float)h1), l2 = (_Float16)(c - (float)h2),
                 l3 = (_Float16)(d - (float)h3);
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  half4 hv = {h0, h1, h2, h3}, lv = {l0, l1, l2, l3};
  unsigned short *p = row + (n >> 5) * 64 + (n & 31);
  *reinterpret_cast<half4 *>(p) = hv;
  *reinterpret_cast<half4 *>(p + 32) = lv;
}
