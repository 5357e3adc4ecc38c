// Streaming-softmax (flash-style) multi-head attention over token windows, exact fp32
// on the CDNA4 matrix cores.
//
// Replaces both attention flavours of the reference's transformer block:
//   * WindowAttention.forward  (vit_nlc.py:219-258): windows of 576 tokens
//     ((24,24) / (12,48) / (48,12)) of the 72x144 token grid, zero-padded bottom/right,
//     padded tokens carrying q = k = v = qkv-bias, UNMASKED softmax;
//   * Attention.forward        (vit_nlc.py:94-112): one 10 368-token "window".
// The reference materialises the score matrix (6.9 GB for a global block, the thing
// that dominates its CPU time); here scores never leave registers.
//
// Mapping to the hardware (wave64, v_mfma_f32_32x32x2_f32):
//   * one wavefront owns 32 queries; a block of NW waves shares the K/V tiles of its
//     (window, head) through LDS.
//   * S^T = K . Q^T is computed (keys x queries), so that in the 32x32 accumulator a lane
//     (q = lane&31, h = lane>>5) holds 16 scores OF ITS OWN QUERY (keys
//     (r&3)+8(r>>2)+4h): the row max / row sum of the online softmax are 16 in-register
//     ops + one cross-half exchange (lane ^ 32) - no LDS round trip.
//   * O^T = V^T . P^T is accumulated (head-dim x queries): the P value a lane holds in
//     register r is exactly the B operand of MFMA step r (its k-slot is the lane's own
//     key), so P never moves; and every O accumulator of a lane belongs to query lane&31,
//     so the online-softmax rescale is a lane-local multiply.
//   * the reduction index of both products may be visited in any order, so the MFMA
//     k-slot h is given the d-range [h*HD/2, (h+1)*HD/2): K fragments are contiguous
//     ds_read_b128 reads from a [32][HD+4] LDS image (pad 4 -> conflict-free), Q lives
//     in HD/2 registers per lane for the whole kernel (pre-multiplied by the softmax
//     scale, like the reference's `q * self.scale`, times log2 e so exp() is v_exp_f32).
//   * K/V tiles of 32 keys are prefetched into VGPRs during the previous tile's 64
//     MFMAs (4096 cycles) and written to LDS between two barriers.
//   * waves whose 32 queries are all padding (bottom half of the lower (48,12) windows)
//     skip the MFMA work but keep staging tiles.
//   * block ids are XCD-remapped so the blocks of one (window, head) share an L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>

#include "../../include/cra5_amd.h"
#include "split.h"

CRA5_RANGE_TU(attn_f32)

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

struct WinGeom {
  int H, W, wh, ww, nwc;  // grid, window, windows per row
};

// window-local token index -> row of the qkv matrix, or -1 for a padded position
__device__ __forceinline__ int token_of(const WinGeom &g, int wr, int wc, int t) {
  const int r = t / g.ww, c = t - r * g.ww;
  const int gr = wr * g.wh + r, gc = wc * g.ww + c;
  return (gr < g.H && gc < g.W) ? gr * g.W + gc : -1;
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void window_attention_f32_kernel(
    const float *__restrict__ qkv, const float *__restrict__ pad_row, float *__restrict__ out,
    unsigned short *__restrict__ out_s, int Kp, int C, int heads, WinGeom g, int q_tiles, float scale) {
  constexpr int HH = HD / 2;                 // d-range per MFMA k-slot
  constexpr int DT = (HD + 31) / 32;         // 32-wide output tiles over the head dim
  constexpr int KS = HD + 4;                 // K LDS row stride (floats)
  constexpr int VS = HD;                     // V LDS row stride
  constexpr int NT = NW * 64;
  constexpr int F4_ROW = HD / 4;             // float4 per K/V row
  constexpr int F4_TILE = 32 * F4_ROW;       // float4 per K (or V) tile
  constexpr int STG = (F4_TILE + NT - 1) / NT;
  static_assert(HD % 8 == 0, "head dim must be a multiple of 8");

  __shared__ __attribute__((aligned(16))) float lds[32 * KS + 32 * VS];
  float *Ks = lds;
  float *Vs = lds + 32 * KS;

  const int L = g.wh * g.ww;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = pid % q_tiles;
  const int wh_id = pid / q_tiles;  // (window, head)
  const int head = wh_id % heads;
  const int win = wh_id / heads;
  const int wr = win / g.nwc, wc = win - wr * g.nwc;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int C3 = 3 * C;
  const int hoff = head * HD;

  // ---- this lane's query ----------------------------------------------------------
  const int tq = (qt * NW + wave) * 32 + l31;
  const int q_tok = (tq < L) ? token_of(g, wr, wc, tq) : -1;
  // wave-uniform: does any lane of this wave own a real query?
  const bool wave_active = __any(q_tok >= 0);
  // a q-tile made only of padding (lower (48,12) windows) has nothing to do at all
  if (!__syncthreads_or(wave_active ? 1 : 0)) return;

  // Scores are kept in the log2 domain: q is pre-multiplied by scale * log2(e) (one
  // rounding per element, like the reference's `q * self.scale`), so the softmax
  // exponential is a bare v_exp_f32 of an exact difference - no |x| * 2^-24 argument
  // error from an extra multiply inside exp().
  const float qs = scale * 1.44269504088896340736f;
  float q[HH];
  {
    const float *qrow = (q_tok >= 0) ? qkv + (size_t)q_tok * C3 : pad_row;
    const float *src = qrow + hoff + h * HH;
#pragma unroll
    for (int i = 0; i < HH / 4; ++i) {
      const float4 v = *reinterpret_cast<const float4 *>(src + 4 * i);
      q[4 * i + 0] = v.x * qs;
      q[4 * i + 1] = v.y * qs;
      q[4 * i + 2] = v.z * qs;
      q[4 * i + 3] = v.w * qs;
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- K/V staging ------------------------------------------------------------------
  float4 sk[STG], sv[STG];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define CRA5_KV_LOAD(J)                                                              \
  {                                                                                  \
    _Pragma("unroll") for (int p = 0; p < STG; ++p) {                                \
      const int idx = tid + p * NT;                                                  \
      sk[p] = zero4;                                                                 \
      sv[p] = zero4;                                                                 \
      if (idx < F4_TILE) {                                                           \
        const int row = idx / F4_ROW, c4 = idx - row * F4_ROW;                       \
        const int kt = (J)*32 + row;                                                 \
        if (kt < L) {                                                                \
          const int tok = token_of(g, wr, wc, kt);                                   \
          const float *base = (tok >= 0) ? qkv + (size_t)tok * C3 : pad_row;         \
          sk[p] = *reinterpret_cast<const float4 *>(base + C + hoff + c4 * 4);       \
          sv[p] = *reinterpret_cast<const float4 *>(base + 2 * C + hoff + c4 * 4);   \
        }                                                                            \
      }                                                                              \
    }                                                                                \
  }
#define CRA5_KV_STORE()                                                              \
  {                                                                                  \
    _Pragma("unroll") for (int p = 0; p < STG; ++p) {                                \
      const int idx = tid + p * NT;                                                  \
      if (idx < F4_TILE) {                                                           \
        const int row = idx / F4_ROW, c4 = idx - row * F4_ROW;                       \
        *reinterpret_cast<float4 *>(Ks + row * KS + c4 * 4) = sk[p];                 \
        *reinterpret_cast<float4 *>(Vs + row * VS + c4 * 4) = sv[p];                 \
      }                                                                              \
    }                                                                                \
  }

  const int n_tiles = (L + 31) / 32;
  CRA5_KV_LOAD(0);
  CRA5_KV_STORE();
  __syncthreads();

  const float *k_base = Ks + l31 * KS + h * HH;

  for (int j = 0; j < n_tiles; ++j) {
    if (j + 1 < n_tiles) CRA5_KV_LOAD(j + 1);

    if (wave_active) {
      // ---- S^T tile: 32 keys x 32 queries --------------------------------------------
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < HH / 4; ++i) {
        const float4 kf = *reinterpret_cast<const float4 *>(k_base + 4 * i);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, q[4 * i + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, q[4 * i + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, q[4 * i + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, q[4 * i + 3], s, 0, 0, 0);
      }
      // keys beyond the window (only when L % 32 != 0, i.e. the 648-token hyper-prior)
      if ((j + 1) * 32 > L) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (key >= L) s[r] = -INFINITY;
        }
      }
      // ---- online softmax (all 16 values belong to query l31) ------------------------
      float mloc = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // first tile: 2^-inf = 0
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
        psum += s[r];
      }
      l_run = l_run * alpha + psum;  // per-half partial sum; halves are merged at the end
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      // ---- O^T += V^T . P^T ------------------------------------------------------------
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int d = 32 * t + l31;
        const bool din = (HD % 32 == 0) || (d < HD);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
          float vf = 0.f;
          if (din) vf = Vs[key * VS + d];
          o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o[t], 0, 0, 0);
        }
      }
    }

    __syncthreads();
    if (j + 1 < n_tiles) {
      CRA5_KV_STORE();
      __syncthreads();
    }
  }

  // ---- normalise + store: lane (q, h) holds O[q][32t + 8g + 4h + 0..3] ---------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);  // merge the two key halves
  if (q_tok >= 0) {
    const float inv = 1.0f / l_tot;
    float *orow = out ? out + (size_t)q_tok * C + hoff : nullptr;
    unsigned short *srow = out_s ? out_s + (size_t)q_tok * 2 * Kp : nullptr;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = 32 * t + 8 * gq + 4 * h;
        if (d < HD) {
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
}

template <int HD, int NW>
int launch(const float *qkv, const float *pad_row, float *out, unsigned short *out_s, int Kp, int C, int heads,
           int H, int W, int wh, int ww, float scale, hipStream_t st) {
  WinGeom g;
  g.H = H;
  g.W = W;
  g.wh = wh;
  g.ww = ww;
  const int nwr = (H + wh - 1) / wh;
  g.nwc = (W + ww - 1) / ww;
  const int L = wh * ww;
  const int q_tiles = (L + NW * 32 - 1) / (NW * 32);
  dim3 grid(q_tiles * nwr * g.nwc * heads), block(NW * 64);
  hipLaunchKernelGGL((window_attention_f32_kernel<HD, NW>), grid, block, 0, st, qkv, pad_row, out, out_s, Kp,
                     C, heads, g, q_tiles, scale);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int cra5_window_attention_f32(const float *qkv, const float *pad_row, float *out, uint16_t *out_split,
                                         int split_kp, int C, int heads, int H, int W, int wh, int ww,
                                         float scale, void *stream) {
  if (!qkv || !pad_row || (!out && !out_split) || heads <= 0 || C % heads) return CRA5_ERR_ARG;
  if (out_split && (split_kp < C || split_kp % 32)) return CRA5_ERR_ARG;
  unsigned short *out_s = out_split;
  const int Kp = split_kp;
  if (wh <= 0 || ww <= 0 || H <= 0 || W <= 0) return CRA5_ERR_ARG;
  if ((C & 3) || ((uintptr_t)qkv & 15) || ((uintptr_t)pad_row & 15) || ((uintptr_t)out & 15)) return CRA5_ERR_ARG;
  const int hd = C / heads;
  hipStream_t st = (hipStream_t)stream;
  const int L = wh * ww;
  if (hd == 64) {
    // 576-token windows: 192 queries per block (3 blocks per window-head, no ragged tail);
    // long sequences: 128 queries per block for finer load balance.
    if (L % 192 == 0 && L <= 1152) return launch<64, 6>(qkv, pad_row, out, out_s, Kp, C, heads, H, W, wh, ww, scale, st);
    return launch<64, 4>(qkv, pad_row, out, out_s, Kp, C, heads, H, W, wh, ww, scale, st);
  }
  if (hd == 72) {
    // hyper-prior (648 tokens, 5 heads): 30 blocks of 4 waves.  Fewer waves per block = more blocks
    // but every block still walks all 21 key tiles, each a serial chain of exact-f32 MFMAs:
    // measured 83 / 88 / 102 us for 4 / 2 / 1 waves (tools/attn_bench.py) - the fix is splitting
    // the KEYS over waves, not the queries.  CRA5_ATT72_NW overrides (1 | 2 | 4).
// variant builds only (tools/build_variant.sh): the product library reads no environment variable
#ifdef CRA5_TUNING_ENV
    static const int forced = [] {
      const char *e = getenv("CRA5_ATT72_NW");
      return e ? atoi(e) : 0;
    }();
#else
    constexpr int forced = 0;
#endif
    const int nwr = (H + wh - 1) / wh, nwc = (W + ww - 1) / ww;
    const long blocks4 = (long)((L + 127) / 128) * nwr * nwc * heads;
    (void)blocks4;
    const int nw = forced ? forced : 4;
    if (nw == 1) return launch<72, 1>(qkv, pad_row, out, out_s, Kp, C, heads, H, W, wh, ww, scale, st);
    if (nw == 2) return launch<72, 2>(qkv, pad_row, out, out_s, Kp, C, heads, H, W, wh, ww, scale, st);
    return launch<72, 4>(qkv, pad_row, out, out_s, Kp, C, heads, H, W, wh, ww, scale, st);
  }
  return CRA5_ERR_ARG;
}
