// Streaming-softmax window / global attention on the f16 matrix cores, fp32-accurate via
// the same hi/lo operand split as gemm_split_f16.hip (3 x v_mfma_f32_32x32x16_f16 per
// product, fp32 accumulate).  Head dim 64, window length a multiple of 32 (576-token
// windows, 10 368-token global attention); other shapes use attention_f32.hip.
//
// Same reference semantics as attention_f32.hip (vit_nlc.py:94-112, 219-258): windows of the
// token grid, bottom/right padding carrying q = k = v = qkv-bias, UNMASKED softmax.
//
// Inputs / outputs are split-f16 matrices: the qkv GEMM epilogue writes [N][3C] split, this
// kernel writes the split [N][C] matrix the proj GEMM reads - no fp32 round trip.
//
// Per wavefront (32 queries), per 32-key tile:
//   S^T = K . Q^T    12 MFMAs (4 k16 steps over d x {lo.hi, hi.lo, hi.hi}); Q fragments live in
//                    32 VGPRs for the whole kernel, K fragments are ds_read_b128 from a
//                    [plane][32 keys][64 d] LDS image (row stride 144 B: conflict-free);
//   softmax          lane (q, h) owns 16 scores of ITS query: log2-domain online softmax
//                    (p = exp2(s*c - m), c = scale*log2 e folded into one FMA), exact skip of
//                    the O rescale when no running max moved in the wave;
//   O^T += V^T . P^T 12 MFMAs: the 16 P values a lane holds ARE its B-operand k-slots (the
//                    reduction index may be visited in any order: slot (t, j) of lane half h is
//                    key (r&3)+8(r>>2)+4h with r = 8t+j), split to hi/lo f16 in registers; V^T
//                    fragments are 2 x ds_read_b64 from a TRANSPOSED [plane][64 d][32 keys] LDS
//                    image (row stride 72 B = 18 dwords: 32 lanes hit 32 distinct even banks).
// 24 MFMAs x 32 cycles = 768 matrix cycles per tile instead of 4096 for the exact-f32 kernel.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/cra5_amd.h"
#include "split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

struct WinGeom {
  int H, W, wh, ww, nwc;
};

__device__ __forceinline__ int token_of(const WinGeom &g, int wr, int wc, int t) {
  if (g.ww == g.W && g.wh == g.H) return t;  // one window = the whole grid (global attention): no div/mod
  const int r = t / g.ww, c = t - r * g.ww;
  const int gr = wr * g.wh + r, gc = wc * g.ww + c;
  return (gr < g.H && gc < g.W) ? gr * g.W + gc : -1;
}

constexpr int HD = 64;
constexpr int KS = 72;   // K LDS row stride (halves): 144 B
constexpr int VS = 36;   // V^T LDS row stride (halves): 72 B

template <int NW, bool HI>
__global__ __launch_bounds__(NW * 64, 2) void window_attention_split_kernel(
    const unsigned short *__restrict__ qkv, long ldq /* halves per row = 2*Kp */,
    const unsigned short *__restrict__ pad_row, float *__restrict__ out, unsigned short *__restrict__ out_s,
    int Kp_out, int C, int heads, WinGeom g, int q_tiles, float scale) {
  constexpr int NT = NW * 64;
  constexpr int PIECES = 32 * 16;                 // 16-byte pieces per K (or V) tile
  constexpr int STG = (PIECES + NT - 1) / NT;

  // [K hi][K lo] : 32 x KS ;  [V^T hi][V^T lo] : 64 x VS
  constexpr int KPL = 32 * KS + 32;   // K plane stride (halves): +64 B so hi/lo planes hit different bank halves
  constexpr int VPL = 64 * VS + 8;    // V^T plane stride: +16 B
#ifdef ATT_UNCOND
#define ATT_IDX(x) min((x), PIECES - 1)
#define ATT_IF(c)
#else
#define ATT_IDX(x) (x)
#define ATT_IF(c) if (c)
#endif
#ifdef ATT_SB
#define ATT_SCHED() __builtin_amdgcn_sched_barrier(0)
#else
#define ATT_SCHED()
#endif
#ifndef ATT_LDS_PAD
#define ATT_LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * KPL + 2 * VPL + ATT_LDS_PAD];
  unsigned short *Ks = lds;
  unsigned short *Vt = lds + 2 * KPL;

  const int L = g.wh * g.ww;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = pid % q_tiles;
  const int wh_id = pid / q_tiles;
  const int head = wh_id % heads;
  const int win = wh_id / heads;
  const int wr = win / g.nwc, wc = win - wr * g.nwc;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int hoff = head * HD;
  // halves offset of this head's q / k / v slice inside a split row (64 d = 2 chunks = 128 halves)
  const long qoff = 2L * hoff, koff = 2L * (C + hoff), voff = 2L * (2 * C + hoff);

  const int tq = (qt * NW + wave) * 32 + l31;
  const int q_tok = (tq < L) ? token_of(g, wr, wc, tq) : -1;
  const bool wave_active = __any(q_tok >= 0);
  if (!__syncthreads_or(wave_active ? 1 : 0)) return;

  // ---- Q fragments: B operand of S^T = K.Q^T; step s covers d = 16s + 8h + (0..7) ---------
  half8 qh[4], ql[4];
  {
    const unsigned short *qrow = ((q_tok >= 0) ? qkv + (size_t)q_tok * ldq : pad_row) + qoff;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int chunk = s >> 1, pp = 2 * (s & 1) + h;
      qh[s] = *reinterpret_cast<const half8 *>(qrow + chunk * 64 + pp * 8);
      ql[s] = *reinterpret_cast<const half8 *>(qrow + chunk * 64 + 32 + pp * 8);
    }
  }
  const float cexp = scale * 1.44269504088896340736f;  // scores -> log2 domain

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  uint4 sk[STG], sv[STG];
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#define CRA5_KV_LOAD(J)                                                                   \
  {                                                                                       \
    _Pragma("unroll") for (int p = 0; p < STG; ++p) {                                     \
      const int idx = ATT_IDX(tid + p * NT);                                             \
      sk[p] = zero4;                                                                      \
      sv[p] = zero4;                                                                      \
      ATT_IF(idx < PIECES) {                                                              \
        /* K: 16 lanes cover one 256-byte row (coalesced).  V: 32 lanes cover 32 keys at the  \
           same 16-byte column, so that the transposed b16 LDS writes of a half-wave land  \
           in 32 consecutive halves of ONE V^T row (bank-conflict-free). */               \
        const int row = idx >> 4, piece = idx & 15;                                       \
        const int vrow = idx & 31, vpiece = idx >> 5;                                     \
        const int tok = token_of(g, wr, wc, (J)*32 + row);                                \
        const int vtok = token_of(g, wr, wc, (J)*32 + vrow);                              \
        const unsigned short *base = (tok >= 0) ? qkv + (size_t)tok * ldq : pad_row;      \
        const unsigned short *vbase = (vtok >= 0) ? qkv + (size_t)vtok * ldq : pad_row;   \
        sk[p] = *reinterpret_cast<const uint4 *>(base + koff + piece * 8);                \
        sv[p] = *reinterpret_cast<const uint4 *>(vbase + voff + vpiece * 8);              \
      }                                                                                   \
    }                                                                                     \
  }
  // piece -> (chunk = piece>>3, plane = (piece>>2)&1, d0 = 32*chunk + 8*(piece&3))
#define CRA5_KV_STORE()                                                                   \
  {                                                                                       \
    _Pragma("unroll") for (int p = 0; p < STG; ++p) {                                     \
      const int idx = tid + p * NT;                                                       \
      if (idx < PIECES) {                                                                 \
        const int row = idx >> 4, piece = idx & 15;                                       \
        const int plane = (piece >> 2) & 1, d0 = 32 * (piece >> 3) + 8 * (piece & 3);     \
        *reinterpret_cast<uint4 *>(Ks + plane * KPL + row * KS + d0) = sk[p];             \
        const int vrow = idx & 31, vpiece = idx >> 5;                                     \
        const int vplane = (vpiece >> 2) & 1, vd0 = 32 * (vpiece >> 3) + 8 * (vpiece & 3);\
        unsigned short *vt = Vt + vplane * VPL + vd0 * VS + vrow;                         \
        const unsigned int w0 = sv[p].x, w1 = sv[p].y, w2 = sv[p].z, w3 = sv[p].w;        \
        vt[0 * VS] = (unsigned short)(w0 & 0xFFFFu);                                      \
        vt[1 * VS] = (unsigned short)(w0 >> 16);                                          \
        vt[2 * VS] = (unsigned short)(w1 & 0xFFFFu);                                      \
        vt[3 * VS] = (unsigned short)(w1 >> 16);                                          \
        vt[4 * VS] = (unsigned short)(w2 & 0xFFFFu);                                      \
        vt[5 * VS] = (unsigned short)(w2 >> 16);                                          \
        vt[6 * VS] = (unsigned short)(w3 & 0xFFFFu);                                      \
        vt[7 * VS] = (unsigned short)(w3 >> 16);                                          \
      }                                                                                   \
    }                                                                                     \
  }

  const int n_tiles = L / 32;
  CRA5_KV_LOAD(0);
  CRA5_KV_STORE();
  __syncthreads();

  const unsigned short *k_base = Ks + l31 * KS + 8 * h;          // + plane*32*KS + 16*s
  const unsigned short *v_base = Vt + l31 * VS + 4 * h;          // + plane*64*VS + 32*dt*VS + 16*t (+8)

  for (int j = 0; j < n_tiles; ++j) {
    if (j + 1 < n_tiles) CRA5_KV_LOAD(j + 1);
    ATT_SCHED();

    if (wave_active) {
      // ---- S^T tile (32 keys x 32 queries) -----------------------------------------------
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#ifdef ATT_SETPRIO
      __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const half8 kh = *reinterpret_cast<const half8 *>(k_base + 16 * st);
        const half8 kl = *reinterpret_cast<const half8 *>(k_base + KPL + 16 * st);
        if (!HI) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], s, 0, 0, 0);
        }
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], s, 0, 0, 0);
      }
      // ---- online softmax, log2 domain ---------------------------------------------------
      float mloc = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * cexp;   // cexp > 0: max commutes with the scale
      const float m_new = fmaxf(m_run, mloc);
      float psum = 0.f;
      half8 ph[2], pl[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], cexp, -m_new));
        psum += p;
        const _Float16 hi = (_Float16)p;
        const _Float16 lo = (_Float16)(p - (float)hi);
        ph[r >> 3][r & 7] = hi;
        pl[r >> 3][r & 7] = lo;
      }
      if (!__all(m_new == m_run)) {   // exact: skip the rescale when no max moved in this wave
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        m_run = m_new;
      }
      l_run += psum;
      // ---- O^T += V^T . P^T ----------------------------------------------------------------
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        half8 vh[2], vl[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned short *vp = v_base + 32 * dt * VS + 16 * t;
          const half4 a0 = *reinterpret_cast<const half4 *>(vp);
          const half4 a1 = *reinterpret_cast<const half4 *>(vp + 8);
          const half4 b0 = *reinterpret_cast<const half4 *>(vp + VPL);
          const half4 b1 = *reinterpret_cast<const half4 *>(vp + VPL + 8);
          vh[dt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          vl[dt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if (!HI) {
          o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[0], ph[t], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[1], ph[t], o[1], 0, 0, 0);
          o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[0], pl[t], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[1], pl[t], o[1], 0, 0, 0);
        }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[0], ph[t], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[1], ph[t], o[1], 0, 0, 0);
      }
    }

    ATT_SCHED();
    __syncthreads();
    if (j + 1 < n_tiles) {
      CRA5_KV_STORE();
      __syncthreads();
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (q_tok >= 0) {
    const float inv = 1.0f / l_tot;
    float *orow = out ? out + (size_t)q_tok * C + hoff : nullptr;
    unsigned short *srow = out_s ? out_s + (size_t)q_tok * 2 * Kp_out : nullptr;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = 32 * t + 8 * gq + 4 * h;
        float4 v;
        v.x = o[t][4 * gq + 0] * inv;
        v.y = o[t][4 * gq + 1] * inv;
        v.z = o[t][4 * gq + 2] * inv;
        v.w = o[t][4 * gq + 3] * inv;
        if (orow) *reinterpret_cast<float4 *>(orow + d) = v;
        if (srow) cra5_store_split4(srow, hoff + d, v.x, v.y, v.z, v.w);
      }
  }
}

template <int NW, bool HI>
int launch(const unsigned short *qkv, long ldq, const unsigned short *pad_row, float *out, unsigned short *out_s,
           int Kp_out, int C, int heads, int H, int W, int wh, int ww, float scale, hipStream_t st) {
  WinGeom g;
  g.H = H;
  g.W = W;
  g.wh = wh;
  g.ww = ww;
  const int nwr = (H + wh - 1) / wh;
  g.nwc = (W + ww - 1) / ww;
  const int L = wh * ww;
  const int q_tiles = (L + NW * 32 - 1) / (NW * 32);
  hipLaunchKernelGGL((window_attention_split_kernel<NW, HI>), dim3(q_tiles * nwr * g.nwc * heads), dim3(NW * 64), 0, st,
                     qkv, ldq, pad_row, out, out_s, Kp_out, C, heads, g, q_tiles, scale);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int cra5_window_attention_split(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split,
                                           float *out, uint16_t *out_split, int out_kp, int C, int heads, int H,
                                           int W, int wh, int ww, float scale, int hi_only, void *stream) {
  if (!qkv_split || !pad_row_split || (!out && !out_split) || heads <= 0 || C % heads) return CRA5_ERR_ARG;
  if (C / heads != 64 || qkv_kp != 3 * C) return CRA5_ERR_ARG;  // head slices must be chunk-aligned
  if (wh <= 0 || ww <= 0 || H <= 0 || W <= 0 || (wh * ww) % 32) return CRA5_ERR_ARG;
  if (out_split && (out_kp < C || out_kp % 32)) return CRA5_ERR_ARG;
  if (((uintptr_t)qkv_split & 15) || ((uintptr_t)pad_row_split & 15)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int L = wh * ww;
  const long ldq = 2L * qkv_kp;
  if (hi_only) {
    if (L % 192 == 0 && L <= 1152)
      return launch<6, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
    return launch<4, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
  }
  if (L % 192 == 0 && L <= 1152)
    return launch<6, false>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
#ifndef ATT_NW_GLOBAL
#define ATT_NW_GLOBAL 4
#endif
  return launch<ATT_NW_GLOBAL, false>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
}
