// fp32 "NT" GEMM on the CDNA4 matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T).
//
// Every dense projection of the VAEformer path goes through this kernel (qkv / proj /
// fc1 / fc2 of the 25+8 transformer blocks, the im2col'ed patch-embed conv, the 1x1
// quant / post-quant convs, the ConvTranspose un-embed as a GEMM, the hyper-prior
// projections).  The reference runs them as ATen fp32 ops (vit_nlc.py:57-59, 96, 111,
// 216-217, 302, 629-632, 741; vaeformer.py:154-155), so the parity path computes in
// exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain and runs at the
// 157 TFLOP/s fp32 rate of gfx950.
//
// Design (gfx950):
//  * 64-wide wavefronts, each owning a (TM x TN) grid of 32x32 accumulator tiles in
//    AGPR/VGPRs; a block is WM x WN waves (default 128x128 tile, 4 waves = 1 per SIMD).
//  * Both operands are K-contiguous, so A and W tiles are staged the same way:
//    128-byte row segments (BK = 32 floats) -> full cache lines per row, float4
//    global loads, ds_write_b128 into a [rows][BK+4] LDS image.  The +4 pad makes
//    the fragment ds_read_b128 conflict-free (row stride 36 dwords -> the 16 lanes of
//    every ds_read_b128 lane group hit 16 distinct 4-dword slots).
//  * The k index inside a BK step is permuted: the MFMA's two k-slots (lane>>5) take
//    k = h*16 + s, so each lane reads 16 CONTIGUOUS floats of its row (4 x b128) for
//    the 16 MFMA steps instead of strided scalars.  Summation order inside a row is
//    therefore k-permuted relative to a naive loop (still one fp32 rounding per
//    product).
//  * Register-prefetch pipeline: tile t+1 is fetched into VGPRs while tile t is
//    consumed from LDS (4096 MFMA cycles per step per wave hide HBM/L2 latency), then
//    written to LDS between two barriers.
//  * XCD-aware tile order: block ids are remapped so that each of the 8 XCDs works on a
//    contiguous band of M-panels (A panel reused out of that XCD's L2 across all N
//    tiles).
//  * Fused epilogues: bias, exact-erf GELU, residual / pos-embed add - no extra
//    elementwise pass over the 42 MB activation.
#include <hip/hip_runtime.h>

#include "../../include/cra5_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;  // floats

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Bijective XCD remap (cdna guide T1): hardware places block b on XCD b % 8; give each
// XCD a contiguous chunk of the logical tile order.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM *WN * 64) void gemm_nt_f32_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw, float *C, int ldc,
    const float *__restrict__ bias, const float *res, int ldr, int M, int N, int K, int flags,
    int tiles_n) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NT = WM * WN * 64;
  constexpr int A_F4 = BM * (BK / 4) / NT;  // float4 per thread per A tile
  constexpr int B_F4 = BN * (BK / 4) / NT;
  static_assert(BM * (BK / 4) % NT == 0 && BN * (BK / 4) % NT == 0, "tile/threads mismatch");

  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDS_STRIDE];
  float *As = lds;
  float *Bs = lds + BM * LDS_STRIDE;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = pid / tiles_n, tn = pid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- staging map: thread -> (row, 16-byte column chunk) ------------------------
  const int c4 = tid % (BK / 4);
  const int r0 = tid / (BK / 4);
  constexpr int ROWS_PER_PASS = NT / (BK / 4);

  float4 ra[A_F4], rb[B_F4];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

#define CRA5_GLOAD(K0)                                                                              \
  {                                                                                                 \
    const int k_ = (K0) + c4 * 4;                                                                   \
    const bool kin_ = k_ < K; /* K % 4 == 0: the float4 is entirely in or out */                    \
    _Pragma("unroll") for (int p = 0; p < A_F4; ++p) {                                              \
      const int r_ = m0 + r0 + p * ROWS_PER_PASS;                                                   \
      ra[p] = zero4;                                                                                \
      if (kin_ && r_ < M) ra[p] = *reinterpret_cast<const float4 *>(A + (size_t)r_ * lda + k_);      \
    }                                                                                               \
    _Pragma("unroll") for (int p = 0; p < B_F4; ++p) {                                              \
      const int r_ = n0 + r0 + p * ROWS_PER_PASS;                                                   \
      rb[p] = zero4;                                                                                \
      if (kin_ && r_ < N) rb[p] = *reinterpret_cast<const float4 *>(W + (size_t)r_ * ldw + k_);      \
    }                                                                                               \
  }
#define CRA5_SSTORE()                                                                               \
  {                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < A_F4; ++p)                                                \
        *reinterpret_cast<float4 *>(As + (r0 + p * ROWS_PER_PASS) * LDS_STRIDE + c4 * 4) = ra[p];   \
    _Pragma("unroll") for (int p = 0; p < B_F4; ++p)                                                \
        *reinterpret_cast<float4 *>(Bs + (r0 + p * ROWS_PER_PASS) * LDS_STRIDE + c4 * 4) = rb[p];   \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float *a_base = As + (wm * TM * 32 + l31) * LDS_STRIDE + h * 16;
  const float *b_base = Bs + (wn * TN * 32 + l31) * LDS_STRIDE + h * 16;

  const int nk = (K + BK - 1) / BK;
  CRA5_GLOAD(0);
  CRA5_SSTORE();
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) CRA5_GLOAD((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(a_base + i * 32 * LDS_STRIDE + kk * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(b_base + j * 32 * LDS_STRIDE + kk * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      CRA5_SSTORE();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds column n = l31, rows (r&3) + 8*(r>>2) + 4*h ----------
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
          float v = acc[i][j][r] + bv;
          if (do_gelu) v = gelu_erf(v);
          if (has_res) v += res[(size_t)m * ldr + n];
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
  }
}

template <int WM, int WN, int TM, int TN>
int launch(const float *A, int lda, const float *W, int ldw, float *C, int ldc, const float *bias,
           const float *res, int ldr, int M, int N, int K, int flags, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
  hipLaunchKernelGGL((gemm_nt_f32_kernel<WM, WN, TM, TN>), grid, block, 0, st, A, lda, W, ldw, C, ldc, bias,
                     res, ldr, M, N, K, flags, tiles_n);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int cra5_gemm_nt_f32(const float *A, int lda, const float *W, int ldw, float *C, int ldc,
                                const float *bias, const float *res, int ldr, int M, int N, int K,
                                int flags, void *stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return CRA5_ERR_ARG;
  if ((K & 3) || (lda & 3) || (ldw & 3)) return CRA5_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_BIAS) && !bias) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_RES) && !res) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // Small problems (hyper-prior: 648 tokens) cannot fill 256 CUs with 128x128 tiles:
  // use 64x64 tiles (4 waves x one 32x32 tile) there.
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles128 < 256) return launch<2, 2, 1, 1>(A, lda, W, ldw, C, ldc, bias, res, ldr, M, N, K, flags, st);
  return launch<2, 2, 2, 2>(A, lda, W, ldw, C, ldc, bias, res, ldr, M, N, K, flags, st);
}
