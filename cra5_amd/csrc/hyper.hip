// Latency-bound kernels of the hyper-prior path (HyperpriorEncoder / HyperpriorDecoder,
// vit_nlc.py:488-551, 696-748): 648 tokens x 360 channels, 0.03 TFLOP per frame, executed
// three times per round trip (h_a once, h_s on the encode AND the decode side) and sitting
// on a frame's critical path between its host entropy-coding phases.  At this size nothing
// is throughput-bound: the big-tile engine (gemm_split_f16.hip) spends its time in
// prologue / pipeline fill / a serial 32-step k-loop on a handful of CUs, and the 32-query
// attention blocks walk all keys serially on 30 CUs (85 us for 0.6 GFLOP).  These kernels
// are built the other way round - many small independent waves, everything straight from
// L2 into registers, no LDS staging, no barriers in the main loops:
//
//   * small_gemm_split_kernel: one wave per 32 x (32*TN) output tile, operands in the
//     split-f16 layout read as MFMA fragments directly from global memory (the whole
//     problem lives in L2 / Infinity Cache), D k-steps of loads in flight per wave; KS
//     waves of a block split the k-steps of one tile (fc2, K = 1440; patch-embed, K = 4096)
//     and are summed in a FIXED order through LDS -> bit-reproducible.  Same arithmetic as
//     the big engine: hi.lo + lo.hi + hi.hi on v_mfma_f32_32x32x16_f16, fp32 accumulate.
//     Epilogue: bias / erf-GELU / residual, fp32 and / or split-f16 output, or the
//     "(p1 p2 c)" un-embed store of HyperpriorDecoder (vit_nlc.py:672-679) fused: with the
//     operands swapped a lane owns one token and 4 consecutive output columns, i.e. one
//     16-byte piece of a row of the [2L][Hp][Wp] parameter image (weight rows are
//     pre-permuted to (c, p1, p2) order by the host) - no [648][8192] intermediate, no
//     pixel-shuffle pass (was: 17x read amplification).
//   * hyper_attention_kernel: exact-fp32 attention on v_mfma_f32_16x16x4_f32 with 16-query
//     tiles (41 x heads blocks instead of 6 x heads) and the KEYS split over the 4 waves of
//     a block (online softmax per wave, merged once through LDS in wave order).  S^T = K.Q^T
//     keeps a lane's scores on its own query, P is the B operand of the PV MFMA in place.
//
// Every reduction order is fixed by the launch geometry, never by timing: h_s must give the
// decoder bit-identical CDF indexes (a flipped index desynchronises the rANS stream).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/cra5_amd.h"
#include "split.h"

CRA5_RANGE_TU(hyper)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// same branch-free erf-GELU as the big engine (gemm_split_f16.hip): the two engines must agree
__device__ __forceinline__ float gelu_erf(float x) {
  const float t = __builtin_fabsf(x) * 0.70710678118654752440f;
  const float k = __builtin_amdgcn_rcpf(__builtin_fmaf(0.4f, t, 1.0f));
  float p = 0.03080804832279682f;
  p = __builtin_fmaf(p, k, -0.3524225652217865f);
  p = __builtin_fmaf(p, k, 1.0205539464950562f);
  p = __builtin_fmaf(p, k, -0.7088391780853271f);
  p = __builtin_fmaf(p, k, 0.6733116507530212f);
  p = __builtin_fmaf(p, k, 0.0958886444568634f);
  p = __builtin_fmaf(p, k, 0.2406993806362152f);
  const float half_erfc = 0.5f * p * k * __builtin_amdgcn_exp2f(-(t * t) * 1.4426950408889634f);
  const float phi = (x >= 0.f) ? 1.0f - half_erfc : half_erfc;
  return x * phi;
}

struct PsGeom {  // un-embed store: token grid (Hz, Wz), patch (p1, 4), output image [Cout][Hz*p1][Wz*4]
  int Hz, Wz, p1, Cout;
};

// ---------------------------------------------------------------------------------------
// small-M split-f16 GEMM
// ---------------------------------------------------------------------------------------
template <int TN, int KS, int D, bool PSHUF>
__global__ __launch_bounds__(KS * 64) void small_gemm_split_kernel(
    const unsigned short *__restrict__ A, long lda, const unsigned short *__restrict__ W, long ldw, float *C, int ldc,
    unsigned short *Cs, long ldcs, const float *__restrict__ bias, const float *res, int ldr, int M, int N, int Kp,
    float wscale_inv, int flags, int tiles_n, PsGeom ps) {
  __shared__ float red[(KS > 1) ? (KS - 1) * TN * 16 * 64 : 1];

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
  const int m0 = tm * 32, n0 = tn * 32 * TN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;

  // fragment sources: one 128-byte chunk [32 hi | 32 lo] per row per k-step; this lane's 16-byte
  // pieces are hi(kk = 0, 1) at halves 8h, 16 + 8h and lo at 32 + 8h, 48 + 8h
  const unsigned short *ap = A + (size_t)min(m0 + l31, M - 1) * lda + 8 * h;
  const unsigned short *wp[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) wp[j] = W + (size_t)min(n0 + 32 * j + l31, N - 1) * ldw + 8 * h;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  half8 fa[D][4], fw[D][TN][4];
#define HY_LOAD(SLOT, STEP)                                                                      \
  {                                                                                              \
    const unsigned short *a_ = ap + (size_t)(STEP)*64;                                           \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) fa[SLOT][c] = *reinterpret_cast<const half8 *>(a_ + 16 * c); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                             \
      const unsigned short *w_ = wp[j] + (size_t)(STEP)*64;                                      \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) fw[SLOT][j][c] = *reinterpret_cast<const half8 *>(w_ + 16 * c); \
    }                                                                                            \
  }
  // pieces: c = 0: hi k 0-15, c = 1: hi k 16-31, c = 2: lo k 0-15, c = 3: lo k 16-31.
  // small terms first (lo.hi, hi.lo, then hi.hi), like the big engine.
#define HY_MMA(X, Y, ACC) (PSHUF ? __builtin_amdgcn_mfma_f32_32x32x16_f16(Y, X, ACC, 0, 0, 0) \
                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(X, Y, ACC, 0, 0, 0))
#define HY_MFMA(SLOT)                                                                            \
  {                                                                                              \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                           \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = HY_MMA(fa[SLOT][2 + kk], fw[SLOT][j][kk], acc[j]); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = HY_MMA(fa[SLOT][kk], fw[SLOT][j][2 + kk], acc[j]); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[j] = HY_MMA(fa[SLOT][kk], fw[SLOT][j][kk], acc[j]);     \
    }                                                                                            \
  }

  const int nk = Kp / 32;
  const int cnt = (nk - wave + KS - 1) / KS;   // this wave's k-steps: wave, wave + KS, ...
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < cnt) HY_LOAD(d, wave + d * KS);
  for (int i = 0; i < cnt; i += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (i + d < cnt) {
        HY_MFMA(d);
        if (i + d + D < cnt) HY_LOAD(d, wave + (i + d + D) * KS);
      }
    }
  }

  // ---- split-K: waves 1..KS-1 park their accumulators, wave 0 adds them in wave order ----------
  if (KS > 1) {
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((wave - 1) * TN + j) * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < KS; ++w)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += red[(((w - 1) * TN + j) * 16 + r) * 64 + lane];
  }

  const bool has_bias = flags & CRA5_EPI_BIAS, do_gelu = flags & CRA5_EPI_GELU, has_res = flags & CRA5_EPI_RES;
  if (PSHUF) {
    // acc[j][r]: column n = n0 + 32 j + 8 (r >> 2) + 4 h + (r & 3), token m = m0 + l31.  Column order is
    // (c, p1, p2): 4 consecutive columns = the 4 horizontal pixels of one (channel, pixel row) -> one float4.
    const int m = m0 + l31;
    if (m >= M) return;
    const int hz = m / ps.Wz, wz = m - hz * ps.Wz;
    const int Hp = ps.Hz * ps.p1, Wp = ps.Wz * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + 32 * j + 8 * g + 4 * h;
        if (n >= N) continue;
        float4 v;
        v.x = acc[j][4 * g + 0] * wscale_inv;
        v.y = acc[j][4 * g + 1] * wscale_inv;
        v.z = acc[j][4 * g + 2] * wscale_inv;
        v.w = acc[j][4 * g + 3] * wscale_inv;
        if (has_bias) {
          v.x += bias[n];
          v.y += bias[n + 1];
          v.z += bias[n + 2];
          v.w += bias[n + 3];
        }
        const int c = n / (4 * ps.p1), p1 = (n >> 2) % ps.p1;
        *reinterpret_cast<float4 *>(C + ((size_t)c * Hp + hz * ps.p1 + p1) * Wp + 4 * wz) = v;
      }
    return;
  }
  // acc[j][r]: row m = m0 + 4 h + (r & 3) + 8 (r >> 2), column n = n0 + 32 j + l31
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + 32 * j + l31;
    const bool nin = n < N;
    const float bv = (has_bias && nin) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 4 * h + (r & 3) + 8 * (r >> 2);
      if (m >= M) continue;
      float v = 0.f;
      if (nin) {
        v = acc[j][r] * wscale_inv + bv;
        if (do_gelu) v = gelu_erf(v);
        if (has_res) v += res[(size_t)m * ldr + n];
        if (C) C[(size_t)m * ldc + n] = v;
      }
      // split output: columns N .. Kp_out are the K padding of the consumer -> zeros
      if (Cs && n < (int)(ldcs >> 1)) cra5_store_split(Cs + (size_t)m * ldcs, n, v);
    }
  }
}

template <int TN, int KS, int D, bool PSHUF>
int launch_small(const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C, int ldc,
                 unsigned short *Cs, long ldcs, const float *bias, const float *res, int ldr, int M, int N, int Kp,
                 float wscale_inv, int flags, PsGeom ps, hipStream_t st) {
  const int tiles_m = (M + 31) / 32;
  int tiles_n = (N + 32 * TN - 1) / (32 * TN);
  if (Cs) tiles_n = ((int)(ldcs >> 1) + 32 * TN - 1) / (32 * TN);   // cover the pad columns too
  hipLaunchKernelGGL((small_gemm_split_kernel<TN, KS, D, PSHUF>), dim3(tiles_m * tiles_n), dim3(KS * 64), 0, st, A,
                     lda, W, ldw, C, ldc, Cs, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, tiles_n, ps);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// exact-fp32 attention over ALL tokens (no windows), 16-query tiles, keys split over waves
// ---------------------------------------------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void hyper_attention_kernel(const float *__restrict__ qkv, float *__restrict__ out,
                                                                   unsigned short *__restrict__ out_s, int Kp,
                                                                   int n_tok, int C, int heads, int n_qt, float scale) {
  constexpr int DS = HD / 4;            // head-dim range of one MFMA k-slot (reduction order is free)
  constexpr int DT = (HD + 15) / 16;    // 16-wide output tiles over the head dim
  static_assert(HD % 8 == 0, "head dim must be a multiple of 8");
  __shared__ float red[(NW - 1) * (DT * 4 + 2) * 64];

  const int qt = blockIdx.x % n_qt, head = blockIdx.x / n_qt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int C3 = 3 * C, hoff = head * HD;

  const int q_tok = qt * 16 + li;
  // log2-domain scores: q * (scale * log2 e), one rounding per element like the reference's q * scale
  const float qs = scale * 1.44269504088896340736f;
  float q[DS];
  {
    const float *src = qkv + (size_t)min(q_tok, n_tok - 1) * C3 + hoff + g * DS;
#pragma unroll
    for (int i = 0; i < DS / 2; ++i) {
      const float2 v = *reinterpret_cast<const float2 *>(src + 2 * i);
      q[2 * i] = v.x * qs;
      q[2 * i + 1] = v.y * qs;
    }
  }
  f32x4 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int n_tiles = (n_tok + 15) / 16;
  float kf[2][DS], vf[2][DT][4];
#define HA_LOAD(B, J)                                                                              \
  {                                                                                                \
    const float *kr = qkv + (size_t)min((J)*16 + li, n_tok - 1) * C3 + C + hoff + g * DS;          \
    _Pragma("unroll") for (int i = 0; i < DS / 2; ++i) {                                           \
      const float2 v = *reinterpret_cast<const float2 *>(kr + 2 * i);                              \
      kf[B][2 * i] = v.x;                                                                          \
      kf[B][2 * i + 1] = v.y;                                                                      \
    }                                                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
      const float *vr = qkv + (size_t)min((J)*16 + 4 * g + r, n_tok - 1) * C3 + 2 * C + hoff;      \
      _Pragma("unroll") for (int t = 0; t < DT; ++t) {                                             \
        const int d = 16 * t + li;                                                                 \
        vf[B][t][r] = ((HD % 16 == 0) || d < HD) ? vr[d] : 0.f;                                    \
      }                                                                                            \
    }                                                                                              \
  }
#define HA_TILE(B, J)                                                                              \
  {                                                                                                \
    f32x4 s = {0.f, 0.f, 0.f, 0.f};                                                                \
    _Pragma("unroll") for (int i = 0; i < DS; ++i) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[B][i], q[i], s, 0, 0, 0); \
    /* s[r] = score of key J*16 + 4g + r for query li */                                           \
    if (((J) + 1) * 16 > n_tok) {                                                                  \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) if ((J)*16 + 4 * g + r >= n_tok) s[r] = -INFINITY; \
    }                                                                                              \
    float mloc = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));                                      \
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));                                                  \
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));                                                  \
    const float m_new = fmaxf(m_run, mloc);                                                        \
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                     \
    float psum = 0.f;                                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                \
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);                                                 \
      psum += s[r];                                                                                \
    }                                                                                              \
    l_run = l_run * alpha + psum;                                                                  \
    m_run = m_new;                                                                                 \
    _Pragma("unroll") for (int t = 0; t < DT; ++t) {                                               \
      o[t] *= alpha;                                                                               \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[B][t][r], s[r], o[t], 0, 0, 0); \
    }                                                                                              \
  }
  // keys of tile j: wave j % NW.  Two register sets: tile j+NW is in flight while tile j computes.
  int j = wave;
  if (j < n_tiles) HA_LOAD(0, j);
  while (j < n_tiles) {
    if (j + NW < n_tiles) HA_LOAD(1, j + NW);
    HA_TILE(0, j);
    j += NW;
    if (j >= n_tiles) break;
    if (j + NW < n_tiles) HA_LOAD(0, j + NW);
    HA_TILE(1, j);
    j += NW;
  }

  // ---- merge the NW partial softmaxes (fixed order: wave 0 + wave 1 + ...) ----------------------
  constexpr int RW = DT * 4 + 2;
  if (wave > 0) {
    float *dst = red + (size_t)(wave - 1) * RW * 64;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(t * 4 + r) * 64 + lane] = o[t][r];
    dst[(DT * 4) * 64 + lane] = m_run;
    dst[(DT * 4 + 1) * 64 + lane] = l_run;
  }
  __syncthreads();
  if (wave > 0) return;
  float m_all = m_run;
#pragma unroll
  for (int w = 1; w < NW; ++w) m_all = fmaxf(m_all, red[((size_t)(w - 1) * RW + DT * 4) * 64 + lane]);
  // a wave that saw no key tile has m = -inf, l = 0, o = 0: its factor is forced to 0 (not 2^(-inf+inf))
  float f0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_all);
  float l_tot = l_run * f0;
#pragma unroll
  for (int t = 0; t < DT; ++t) o[t] *= f0;
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    const float *src = red + (size_t)(w - 1) * RW * 64;
    const float mw = src[(DT * 4) * 64 + lane];
    const float fw_ = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - m_all);
    l_tot += src[(DT * 4 + 1) * 64 + lane] * fw_;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t][r] += src[(t * 4 + r) * 64 + lane] * fw_;
  }
  // l is a per-lane partial over the lane's own key slots: join the 4 slot groups of a query
  l_tot += __shfl_xor(l_tot, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (q_tok < n_tok) {
    const float inv = 1.0f / l_tot;
    float *orow = out ? out + (size_t)q_tok * C + hoff : nullptr;
    unsigned short *srow = out_s ? out_s + (size_t)q_tok * 2 * Kp : nullptr;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = 16 * t + 4 * g;   // o[t][r]: head-dim d + r of query li
      if (d < HD) {
        const float4 v = make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv);
        if (orow) *reinterpret_cast<float4 *>(orow + d) = v;
        if (srow) cra5_store_split4(srow, hoff + d, v.x, v.y, v.z, v.w);
      }
    }
  }
}

}  // namespace

extern "C" int cra5_internal_gemm_split_tile(const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C,
                                             int ldc, unsigned short *Cs, long ldcs, const float *bias,
                                             const float *res, int ldr, int M, int N, int Kp, float wscale_inv,
                                             int flags, int tile, hipStream_t st);

extern "C" int cra5_small_gemm_nt_split(const uint16_t *A, int lda_kp, const uint16_t *W, int ldw_kp, float *C, int ldc,
                                        uint16_t *C_split, int ldc_split_kp, const float *bias, const float *res,
                                        int ldr, int M, int N, int Kp, float wscale_inv, int flags, int ps_Hz, int ps_Wz,
                                        int ps_p1, int ps_p2, void *stream) {
  if (!A || !W || (!C && !C_split) || M <= 0 || N <= 0 || Kp <= 0 || (Kp % 32)) return CRA5_ERR_ARG;
  if (lda_kp < Kp || ldw_kp < Kp || (lda_kp % 32) || (ldw_kp % 32)) return CRA5_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_BIAS) && !bias) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_RES) && !res) return CRA5_ERR_ARG;
  if (C_split && (ldc_split_kp % 32 || ldc_split_kp < N)) return CRA5_ERR_ARG;
  if (flags & CRA5_GEMM_HI_ONLY) return CRA5_ERR_ARG;   // the hyper-prior path is always fp32-accurate
  hipStream_t st = (hipStream_t)stream;
  const long lda = 2L * lda_kp, ldw = 2L * ldw_kp, ldcs = 2L * ldc_split_kp;
  PsGeom ps{ps_Hz, ps_Wz, ps_p1, 0};
  if (ps_Hz > 0) {
    // un-embed store: N = Cout * p1 * 4 columns in (c, p1, p2) order, C = image [Cout][Hz*p1][Wz*4]
    if (ps_p2 != 4 || ps_p1 <= 0 || M != ps_Hz * ps_Wz || (N % (4 * ps_p1)) || !C || C_split || ((uintptr_t)C & 15) ||
        (flags & (CRA5_EPI_GELU | CRA5_EPI_RES)))
      return CRA5_ERR_ARG;
    ps.Cout = N / (4 * ps_p1);
    return launch_small<4, 1, 2, true>(A, lda, W, ldw, C, 0, nullptr, 0, bias, nullptr, 0, M, N, Kp, wscale_inv, flags,
                                       ps, st);
  }
#define HY_GO(TN, KS, D) \
  return launch_small<TN, KS, D, false>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, ps, st)
  const long tiles1 = (long)((M + 31) / 32) * ((N + 31) / 32);
  const int nk = Kp / 32;
  // The instantiation is a function of the SHAPE only: it fixes the split-K factor and with it the fp32 summation
  // order of h_s, which encoder and decoder must share bit for bit.  The sweep override (CRA5_HY_GEMM = "TN KS",
  // tools/hyper_gemm_sweep.py) exists only in -DCRA5_HY_SWEEP builds (tools/build_variant.sh), never in the product.
#ifdef CRA5_HY_SWEEP
  static const int forced = [] {
    const char *e = getenv("CRA5_HY_GEMM");
    int tn = 0, ks = 0;
    if (e && sscanf(e, "%d %d", &tn, &ks) == 2) return tn * 100 + ks;
    return 0;
  }();
  switch (forced) {
    case 101: HY_GO(1, 1, 4);
    case 102: HY_GO(1, 2, 4);
    case 104: HY_GO(1, 4, 2);
    case 108: HY_GO(1, 8, 2);
    case 201: HY_GO(2, 1, 3);
    case 202: HY_GO(2, 2, 2);
    case 204: HY_GO(2, 4, 2);
    case 401: HY_GO(4, 1, 2);
    default: break;
  }
#endif
  // Measured on MI355X.  Warm micro-benchmark per shape (tools/hyper_gemm_sweep.sh, kernel us, this kernel vs the
  // LDS-tiled engine at a fixed 64 / 128 tile): 648x360x4096 34 vs 65, 648x360x1440 15.1 vs 16.2, 648x1080x360
  // 13.2 vs 8.6, 648x1440x360 16.1 vs 9.1, 648x8192x360 60 vs 26.  IN SITU (every GEMM of h_s meets weights that
  // are cold in L2; tools/hyper_bench.py) the picture flips for the short reductions too - h_s 436 us with this
  // kernel everywhere vs 507 us with the 64-tile engine forwarded for K = 360: four k-steps of loads in flight
  // per wave ride out the misses that a two-stage LDS pipeline stalls on.  So: this kernel for every shape.
  if (nk >= 96) HY_GO(1, 8, 2);
  if (nk >= 32 && tiles1 <= 1024) HY_GO(1, 4, 2);
  if (tiles1 > 4096) HY_GO(4, 1, 2);
  if (tiles1 > 1536) HY_GO(2, 1, 3);
  HY_GO(1, 1, 4);
#undef HY_GO
}

extern "C" int cra5_hyper_attention_f32(const float *qkv, float *out, uint16_t *out_split, int split_kp, int n_tok,
                                        int C, int heads, float scale, void *stream) {
  if (!qkv || (!out && !out_split) || n_tok <= 0 || heads <= 0 || C % heads) return CRA5_ERR_ARG;
  if (out_split && (split_kp < C || split_kp % 32)) return CRA5_ERR_ARG;
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || (C & 3)) return CRA5_ERR_ARG;
  const int hd = C / heads;
  const int n_qt = (n_tok + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (hd == 72) {
    hipLaunchKernelGGL((hyper_attention_kernel<72, 4>), dim3(n_qt * heads), dim3(256), 0, st, qkv, out, out_split,
                       split_kp, n_tok, C, heads, n_qt, scale);
    return (int)hipGetLastError();
  }
  if (hd == 64) {
    hipLaunchKernelGGL((hyper_attention_kernel<64, 4>), dim3(n_qt * heads), dim3(256), 0, st, qkv, out, out_split,
                       split_kp, n_tok, C, heads, n_qt, scale);
    return (int)hipGetLastError();
  }
  return CRA5_ERR_ARG;
}
