// fp32-accurate "NT" GEMM on the f16 matrix cores via operand splitting:
//     C[M,N] = epi( wscale_inv * A[M,K] . W[N,K]^T )
//
// gfx950 has no TF32/xf32 path and its exact f32 MFMA runs at the f32 VECTOR rate
// (157 TFLOP/s); v_mfma_f32_32x32x16_f16 is 16x faster.  Each fp32 operand x is stored as
// two halves  x = hi + lo,  hi = f16(x), lo = f16(x - hi)   (22 significant bits), and
//     a.b  ~=  hi_a.hi_b + hi_a.lo_b + lo_a.hi_b          (3 MFMAs, fp32 accumulate)
// drops only the lo.lo term (2^-22 relative).  f16 x f16 products are exact in fp32 and the
// MFMA reduces 16 products per accumulate step, so the fp32 accumulation chain is K/16 long
// instead of K: measured against float64 this is MORE accurate than the sequential-fmaf
// chain of the exact-f32 MFMA kernel (gemm_f32.hip) for the K = 1024..29 480 reductions of
// this model, at ~4x its speed.  Range: |x| < 65 504 per element; values below 2^-3 carry
// an ABSOLUTE error <= 2^-25 (fine for O(1) activations); weights (O(0.02)) are scaled by a
// per-tensor power of two at split time and `wscale_inv` undoes it exactly in the epilogue.
//
// "Split-f16" matrix layout (same bytes and row stride as fp32): per row, K is cut into
// chunks of 32; chunk c occupies 128 contiguous bytes = 32 hi halves then 32 lo halves.
// One 128-byte line therefore feeds one BK = 32 step of BOTH planes of a row: coalesced
// full-line global loads, and the producers (LayerNorm, GELU epilogue, attention, patch
// gather) write their output directly in this format - no conversion pass.
//
// Kernel structure (per 128x128 tile, 4 waves = one per SIMD, each a 2x2 grid of 32x32
// accumulators): register-prefetch of the next 4 plane tiles (A hi/lo, W hi/lo) while the
// current ones are consumed from LDS; LDS rows padded 64 -> 80 bytes so that every
// ds_read_b128 lane group touches 16 distinct 16-byte slots; 24 MFMAs per wave per step,
// issued plane-major (4 independent accumulators between dependent MFMAs); XCD-aware
// tile order; fused bias / exact-erf GELU / residual epilogue with fp32 and/or split-f16
// output.  Long reductions (K > 8192: the patch-embed conv, K = 29 480) additionally flush
// the MFMA accumulators into a second fp32 accumulator set every 512 k (two-level sum).
#include <hip/hip_runtime.h>

#include "../../include/cra5_amd.h"
#include "split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;         // elements per k-step
constexpr int ROW_H = 40;      // LDS row stride in halves (64 B data + 16 B pad)

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

template <int WM, int WN, int TM, int TN, bool LONGK>
__global__ __launch_bounds__(WM *WN * 64) void gemm_nt_split_kernel(
    const unsigned short *__restrict__ A, long lda, const unsigned short *__restrict__ W, long ldw, float *C,
    int ldc, unsigned short *Cs, long ldcs, const float *__restrict__ bias, const float *res, int ldr, int M,
    int N, int Kp, float wscale_inv, int flags, int tiles_n) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NT = WM * WN * 64;
  constexpr int ROWS_PER_PASS = NT / 8;  // 8 x 16 B per row per k-step (hi 64 B | lo 64 B)
  constexpr int A_P = BM / ROWS_PER_PASS;
  constexpr int B_P = BN / ROWS_PER_PASS;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/threads mismatch");

  // [A hi][A lo][W hi][W lo], each rows x ROW_H halves
  __shared__ __attribute__((aligned(16))) unsigned short lds[(2 * BM + 2 * BN) * ROW_H];
  unsigned short *As = lds;                  // plane p at As + p * BM * ROW_H
  unsigned short *Bs = lds + 2 * BM * ROW_H;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = pid / tiles_n, tn = pid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  const int c8 = tid & 7;       // 16-byte piece of the 128-byte chunk: 0-3 hi, 4-7 lo
  const int r0 = tid >> 3;
  const int plane = c8 >> 2, pc = c8 & 3;

  uint4 ra[A_P], rb[B_P];
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

#define CRA5_GLOAD(KT)                                                                         \
  {                                                                                            \
    _Pragma("unroll") for (int p = 0; p < A_P; ++p) {                                          \
      const int r_ = m0 + r0 + p * ROWS_PER_PASS;                                              \
      ra[p] = zero4;                                                                           \
      if (r_ < M) ra[p] = *reinterpret_cast<const uint4 *>(A + (size_t)r_ * lda + (size_t)(KT)*64 + c8 * 8); \
    }                                                                                          \
    _Pragma("unroll") for (int p = 0; p < B_P; ++p) {                                          \
      const int r_ = n0 + r0 + p * ROWS_PER_PASS;                                              \
      rb[p] = zero4;                                                                           \
      if (r_ < N) rb[p] = *reinterpret_cast<const uint4 *>(W + (size_t)r_ * ldw + (size_t)(KT)*64 + c8 * 8); \
    }                                                                                          \
  }
#define CRA5_SSTORE()                                                                          \
  {                                                                                            \
    _Pragma("unroll") for (int p = 0; p < A_P; ++p)                                            \
        *reinterpret_cast<uint4 *>(As + plane * BM * ROW_H + (r0 + p * ROWS_PER_PASS) * ROW_H + pc * 8) = ra[p]; \
    _Pragma("unroll") for (int p = 0; p < B_P; ++p)                                            \
        *reinterpret_cast<uint4 *>(Bs + plane * BN * ROW_H + (r0 + p * ROWS_PER_PASS) * ROW_H + pc * 8) = rb[p]; \
  }

  f32x16 acc[TM][TN];
  f32x16 master[LONGK ? TM : 1][LONGK ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if (LONGK) master[i][j][r] = 0.f;
      }

  const unsigned short *a_base = As + (wm * TM * 32 + l31) * ROW_H + h * 8;
  const unsigned short *b_base = Bs + (wn * TN * 32 + l31) * ROW_H + h * 8;

  const int nk = Kp / BK;
  CRA5_GLOAD(0);
  CRA5_SSTORE();
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) CRA5_GLOAD(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const half8 *>(a_base + i * 32 * ROW_H + kk * 16);
        al[i] = *reinterpret_cast<const half8 *>(a_base + BM * ROW_H + i * 32 * ROW_H + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const half8 *>(b_base + j * 32 * ROW_H + kk * 16);
        bl[j] = *reinterpret_cast<const half8 *>(b_base + BN * ROW_H + j * 32 * ROW_H + kk * 16);
      }
      // small terms first, plane-major: TM*TN independent accumulators between dependent MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
    if (LONGK && ((kt & 15) == 15)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            master[i][j][r] += acc[i][j][r];
            acc[i][j][r] = 0.f;
          }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      CRA5_SSTORE();
      __syncthreads();
    }
  }

  const bool has_bias = flags & CRA5_EPI_BIAS;
  const bool do_gelu = flags & CRA5_EPI_GELU;
  const bool has_res = flags & CRA5_EPI_RES;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + l31;
    if (n >= N) continue;
    const float bv = has_bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + (wm * TM + i) * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < M) {
          float v = acc[i][j][r];
          if (LONGK) v += master[i][j][r];
          v = v * wscale_inv + bv;
          if (do_gelu) v = gelu_erf(v);
          if (has_res) v += res[(size_t)m * ldr + n];
          if (C) C[(size_t)m * ldc + n] = v;
          if (Cs) cra5_store_split(Cs + (size_t)m * ldcs, n, v);
        }
      }
    }
  }
}

// fp32 [rows][K] (row stride ldx) -> split-f16 [rows][2*Kp] halves, x * scale, pad zeros.
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, long ldx,
                                                         unsigned short *__restrict__ out, int rows, int K, int Kp,
                                                         float scale) {
  const size_t total = (size_t)rows * Kp;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / Kp), k = (int)(e - (size_t)r * Kp);
    const float v = (k < K) ? x[(size_t)r * ldx + k] * scale : 0.f;
    cra5_store_split(out + (size_t)r * 2 * Kp, k, v);
  }
}

template <int WM, int WN, int TM, int TN, bool LONGK>
int launch(const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C, int ldc,
           unsigned short *Cs, long ldcs, const float *bias, const float *res, int ldr, int M, int N, int Kp,
           float wscale_inv, int flags, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, LONGK>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0,
                     st, A, lda, W, ldw, C, ldc, Cs, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, tiles_n);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int cra5_gemm_nt_split(const uint16_t *A, const uint16_t *W, float *C, int ldc, uint16_t *C_split,
                                  int ldc_split_kp, const float *bias, const float *res, int ldr, int M, int N,
                                  int Kp, float wscale_inv, int flags, void *stream) {
  if (!A || !W || (!C && !C_split) || M <= 0 || N <= 0 || Kp <= 0 || (Kp % BK)) return CRA5_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_BIAS) && !bias) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_RES) && !res) return CRA5_ERR_ARG;
  if (C_split && (ldc_split_kp % 32 || ldc_split_kp < N)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long ld = 2L * Kp, ldcs = 2L * ldc_split_kp;
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (Kp > 8192)
    return launch<2, 2, 2, 2, true>(A, ld, W, ld, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
  if (tiles128 < 256)
    return launch<2, 2, 1, 1, false>(A, ld, W, ld, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
  return launch<2, 2, 2, 2, false>(A, ld, W, ld, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
}

extern "C" int cra5_split_f16(const float *x, int ldx, uint16_t *out, int rows, int K, int Kp, float scale,
                              void *stream) {
  if (!x || !out || rows <= 0 || K <= 0 || Kp < K || (Kp % 32)) return CRA5_ERR_ARG;
  size_t total = (size_t)rows * Kp;
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, out, rows,
                     K, Kp, scale);
  return (int)hipGetLastError();
}
