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
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
// hardware transpose read (gfx950): 64 bits per lane, 16-bit elements exchanged inside 16-lane groups
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_TR_READ(P) \
  __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4 *)(P)))
#else
#define CRA5_TR_READ(P) (*reinterpret_cast<const half4 *>(P))
#endif

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
constexpr int KS = 64;   // K and V LDS row stride (halves): 128 B, unpadded (LDS-DMA writes are linear),
constexpr int VS = 64;   // bank conflicts avoided by XOR swizzles of the piece index (see the kernel)
constexpr int MAX_WIN_TOKENS = 1536;   // windowed (not whole-grid) launches: L <= this (12 KB offset table)

template <int NW, bool HI, bool GLOBAL>
__global__ __launch_bounds__(NW * 64, (NW >= 8 ? 1 : 2)) void window_attention_split_kernel(
    const unsigned short *__restrict__ qkv, long ldq /* halves per row = 2*Kp */,
    const unsigned short *__restrict__ pad_row, float *__restrict__ out, unsigned short *__restrict__ out_s,
    int Kp_out, int C, int heads, WinGeom g, int q_tiles, float scale) {
  constexpr int NT = NW * 64;
  // [K hi][K lo] : 32 x 128 B ;  [V hi][V lo] : 32 x 128 B (row-major like K, transposed by the READ)
  constexpr int KPL = 32 * KS;        // K plane stride (halves)
  constexpr int VPL = 32 * VS;        // V plane stride
#ifndef ATT_LDS_PAD
#define ATT_LDS_PAD 0
#endif
#ifndef ATT_VALU_PER_MFMA
#define ATT_VALU_PER_MFMA 12
#endif
  // two K buffers and two V^T buffers: tile j+1's scores are issued to the matrix pipe BEFORE
  // the softmax of tile j, so K runs one tile ahead of V; one barrier per key tile.
  constexpr int KBUF = 2 * KPL, VBUF = 2 * VPL;
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * KBUF + 2 * VBUF + ATT_LDS_PAD];
  unsigned short *Ks = lds;
  unsigned short *Vt = lds + 2 * KBUF;
  // windowed launches: byte offset (from qkv) of every window token's row, pad tokens -> the pad
  // row; built once per block so that the per-tile staging needs no division / multiply.
  constexpr int TAB = GLOBAL ? 1 : MAX_WIN_TOKENS;
  __shared__ long long tab[TAB];

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
  if (!GLOBAL) {
    const long long pad_delta = reinterpret_cast<const char *>(pad_row) - reinterpret_cast<const char *>(qkv);
    for (int t = tid; t < L; t += NT) {
      const int tok = token_of(g, wr, wc, t);
      tab[t] = (tok >= 0) ? (long long)tok * ldq * 2 : pad_delta;
    }
  }
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

  const int n_tiles = L / 32;
  // ---- staging by LDS-DMA (global_load_lds_dwordx4): a K or V tile is 2 planes x 32 rows x 128 B =
  // 8 one-KB instructions (8 rows x 128 B each); the 16 instructions of a (K tile, V tile) pair are
  // dealt round-robin to the block's waves.  No staging VGPRs, no ds_write pass, and a full tile
  // iteration between issue and use.  The destination is linear (row = lane/8, physical 16-byte piece
  // = lane%8), so the bank swizzles are applied to the per-lane SOURCE address:
  //   K: piece p of row r at p ^ ((r >> 1) & 7)        (conflict-free ds_read_b128 fragments)
  //   V: piece p of row r at p ^ (((r >> 1) & 1) << 2) (conflict-free ds_read_b64_tr_b16, below)
  // Source of (row, plane, piece p): the head's 64-d slice of a split row = 2 chunks [32 hi | 32 lo]:
  // halves offset (p >> 2) * 64 + plane * 32 + (p & 3) * 8.
  constexpr int NINS = 16;                                  // DMA instructions per (K tile, V tile)
  constexpr int IPW = (NINS + NW - 1) / NW;                 // per wave
  const int lrow8 = lane >> 3, lp8 = lane & 7;
  long src_off[IPW];                                        // halves, inside a split row
  int src_row[IPW];                                         // row of the tile this lane fetches
#pragma unroll
  for (int q = 0; q < IPW; ++q) {
    const int id = min(wave + q * NW, NINS - 1);            // 0-7: K, 8-15: V; (id >> 2) & 1: plane; id & 3: 8-row group
    const bool isV = id >= 8;
    const int plane = (id >> 2) & 1, row = (id & 3) * 8 + lrow8;
    const int p = isV ? (lp8 ^ (((row >> 1) & 1) << 2)) : (lp8 ^ ((row >> 1) & 7));
    src_row[q] = row;
    src_off[q] = (isV ? voff : koff) + (p >> 2) * 64 + plane * 32 + (p & 3) * 8;
  }
  // whole-grid launches walk running row pointers (+32 rows per tile and per call: K and V tiles are
  // requested in order 0, 1, 2, ...); windowed ones look the rows up in the table.
  const unsigned short *gq[IPW];
#pragma unroll
  for (int q = 0; q < IPW; ++q) gq[q] = qkv + (size_t)src_row[q] * ldq + src_off[q];
  const long tile_step = 32 * ldq;
#define CRA5_ROW(JJ, ROW) \
  reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(qkv) + tab[(JJ)*32 + (ROW)])
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_GLDS16(SRC, DST) __builtin_amdgcn_global_load_lds(SRC, DST, 16, 0, 0)
#else
#define CRA5_GLDS16(SRC, DST) (void)(SRC)
#endif
  // request K tile JK into K buffer KBF and / or V tile JV into V buffer VBF (WANT_K / WANT_V: wave-uniform)
#define CRA5_STAGE(JK, KBF, WANT_K, JV, VBF, WANT_V)                                      \
  {                                                                                       \
    _Pragma("unroll") for (int q = 0; q < IPW; ++q) {                                     \
      const int id = wave + q * NW;                                                       \
      if (id < NINS) {                                                                    \
        const bool isV = id >= 8;                                                         \
        const bool want = isV ? (WANT_V) : (WANT_K);                                      \
        if (want) {                                                                       \
          const unsigned short *sp_;                                                      \
          if (GLOBAL) {                                                                   \
            sp_ = gq[q];                                                                  \
            gq[q] += tile_step;                                                           \
          } else {                                                                        \
            sp_ = CRA5_ROW(isV ? (JV) : (JK), src_row[q]) + src_off[q];                   \
          }                                                                               \
          unsigned short *dst_ = isV ? Vt + (VBF)*VBUF + (id - 8) * 512 : Ks + (KBF)*KBUF + id * 512; \
          CRA5_GLDS16(sp_, dst_);                                                         \
        }                                                                                 \
      }                                                                                   \
    }                                                                                     \
  }

  // S^T tile (32 keys x 32 queries) of the K buffer KB: 12 MFMAs on ATT_S_CHAINS independent
  // accumulators (a dependent 32x32x16 MFMA cannot issue back-to-back), summed at the end.
#ifndef ATT_S_CHAINS
#define ATT_S_CHAINS 1
#endif
#define ATT_KFRAG(PTR, ALT) (*reinterpret_cast<const half8 *>(PTR))
#define CRA5_SCORES(DST, KB)                                                              \
  {                                                                                       \
    f32x16 acc_[ATT_S_CHAINS];                                                            \
    _Pragma("unroll") for (int c = 0; c < ATT_S_CHAINS; ++c)                              \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc_[c][r] = 0.f;                    \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                    \
      const half8 kh = ATT_KFRAG(k_base + (KB)*KBUF + kp_off[st], qh[3 - st]);            \
      const half8 kl = ATT_KFRAG(k_base + (KB)*KBUF + KPL + kp_off[st], ql[3 - st]);      \
      if (!HI) {                                                                          \
        acc_[(3 * st) % ATT_S_CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], acc_[(3 * st) % ATT_S_CHAINS], 0, 0, 0); \
        acc_[(3 * st + 1) % ATT_S_CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], acc_[(3 * st + 1) % ATT_S_CHAINS], 0, 0, 0); \
      }                                                                                   \
      acc_[(3 * st + 2) % ATT_S_CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], acc_[(3 * st + 2) % ATT_S_CHAINS], 0, 0, 0); \
    }                                                                                     \
    _Pragma("unroll") for (int c = 1; c < ATT_S_CHAINS; ++c)                              \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc_[0][r] += acc_[c][r];            \
    DST = acc_[0];                                                                        \
  }

  // K fragment of k16-step st: logical piece 2 st + h of key row l31, at physical piece ^ ((l31 >> 1) & 7)
  const unsigned short *k_base = Ks + l31 * KS;                  // + buf*KBUF + plane*KPL + kp_off[st]
  int kp_off[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) kp_off[st] = ((2 * st + h) ^ ((l31 >> 1) & 7)) * 8;
  // V^T fragments (A operand: row d = l31, 8 keys per lane) come out of the ROW-MAJOR V image through
  // ds_read_b64_tr_b16: inside each 16-lane group, lane l' receives element (l' & 3) of the 8-byte
  // slots addressed by lanes (l' >> 2) + {0, 4, 8, 12} (probed: tools/probes/tr_probe.hip).  Lane l'
  // therefore ADDRESSES V[kbase + (l' >> 2)][d0 + 4 (l' & 3) ..+3] and RECEIVES V[kbase + 0..3][d0 + l'],
  // d0 = 32 dt + 16 ((lane >> 4) & 1): four consecutive keys of its own d.  With 128-byte rows, key rows
  // r and r + 2 share banks: the 16-byte piece index is XORed with ((key >> 1) & 1) << 2, i.e. the two
  // 32-d halves of rows 2, 3 (mod 4) are swapped, and the four key rows x two d-halves of a 32-lane LDS
  // cycle fall on 8 disjoint 8-bank ranges.  (key >> 1) & 1 = (l' >> 3) & 1 for every key base used.
  const int vsk = (lane >> 3) & 1;
  const unsigned short *v_base0 = Vt + (4 * h + ((lane & 15) >> 2)) * VS + 32 * vsk + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned short *v_base1 = Vt + (4 * h + ((lane & 15) >> 2)) * VS + 32 * (vsk ^ 1) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // v_base{dt} + buf*VBUF + plane*VPL + (16*t + 8*a)*VS

  // cross-half max: lanes l and l+32 own the two halves of one query's 32 scores
#define CRA5_XHALF_MAX(X)                                                                 \
  ({                                                                                      \
    float a_ = (X), b_ = (X);                                                             \
    asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a_), "+v"(b_));              \
    fmaxf(a_, b_);                                                                        \
  })
#define CRA5_TILE_MAX(S)                                                                  \
  ({                                                                                      \
    float m_ = fmaxf(fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3])), fmaxf(fmaxf(S[4], S[5]), fmaxf(S[6], S[7]))); \
    float n_ = fmaxf(fmaxf(fmaxf(S[8], S[9]), fmaxf(S[10], S[11])), fmaxf(fmaxf(S[12], S[13]), fmaxf(S[14], S[15]))); \
    CRA5_XHALF_MAX(fmaxf(m_, n_)) * cexp; /* cexp > 0: max commutes with the scale */      \
  })

  // prologue: K(0), V(0), K(1) resident; S(0) done.  (K tiles are requested in order 0, 1, 2, ... and V
  // tiles likewise: the running pointers of the whole-grid case rely on it.)
  CRA5_STAGE(0, 0, true, 0, 0, true);
  CRA5_STAGE(1, 1, n_tiles > 1, 0, 0, false);
  __syncthreads();   // (hipcc drains this wave's LDS-DMA - vmcnt - before the barrier)
  f32x16 s_cur;
  CRA5_SCORES(s_cur, 0);
  float mloc = CRA5_TILE_MAX(s_cur);
  // iteration 0 re-fills K buffer 0 with tile 2: every wave must be done reading tile 0 from it
  __syncthreads();

  // Waves past the end of the window (q_tok < 0 for all lanes) run the same instruction stream
  // on the pad row and store nothing: one code path, no divergent barriers.
  for (int j = 0; j < n_tiles; ++j) {
    const int kb = (j + 1) & 1, vb = j & 1;
#ifndef ATT_SKIP_STAGE
    // K(j+2) -> the buffer tile j's scores came from, V(j+1) -> the buffer tile j-1's PV read: both were
    // last read before the barrier that ended the previous iteration; they land during this one and
    // become visible at its barrier.
    CRA5_STAGE(j + 2, j & 1, j + 2 < n_tiles, j + 1, (j + 1) & 1, j + 1 < n_tiles);
#endif
    // tile j's running max is known before its softmax starts (mloc was reduced in the shadow of
    // the previous tile's PV MFMAs), so the rare O rescale sits at the top and everything below
    // is ONE basic block the scheduler can interleave.
    const float m_new = fmaxf(m_run, mloc);
    if (!__all(m_new == m_run)) {   // exact: skip the rescale when no max moved in this wave
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    // ---- tile j+1's 12 score MFMAs, interleaved with tile j's softmax VALU work
    // (past the last tile the scores of a stale K buffer are computed and discarded.)
    f32x16 s_next;
#ifdef ATT_SKIP_S
    s_next = s_cur;
#else
    CRA5_SCORES(s_next, kb);
#endif
    float psum = 0.f;
    half8 ph[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#if defined(ATT_SKIP_SOFTMAX)
      const float p = s_cur[r];
      ph[r >> 3][r & 7] = __builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, p));
      pl[r >> 3][r & 7] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, p) >> 16));
      psum += p;
#else
      const float p = __builtin_amdgcn_exp2f(fmaf(s_cur[r], cexp, -m_new));
      psum += p;
      const _Float16 hi = (_Float16)p;
      const _Float16 lo = (_Float16)(p - (float)hi);
      ph[r >> 3][r & 7] = hi;
      pl[r >> 3][r & 7] = lo;
#endif
    }
    l_run += psum;
    // ---- O^T += V^T . P^T, with the staging traffic and tile j+1's max in its shadow
#ifdef ATT_SKIP_PV
    o[0][0] += (float)ph[0][0] + (float)pl[1][7] + (float)ph[1][3] + (float)pl[0][5];
#else
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      half8 vh[2], vl[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned short *vp = (dt ? v_base1 : v_base0) + vb * VBUF + 16 * t * VS;
        const half4 a0 = CRA5_TR_READ(vp);
        const half4 a1 = CRA5_TR_READ(vp + 8 * VS);
        const half4 b0 = CRA5_TR_READ(vp + VPL);
        const half4 b1 = CRA5_TR_READ(vp + VPL + 8 * VS);
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
#endif
    s_cur = s_next;
    mloc = CRA5_TILE_MAX(s_cur);
#ifdef ATT_SGB
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the 8 K-fragment ds_reads first
#pragma unroll
    for (int i = 0; i < (HI ? 4 : 12); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one score MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, ATT_VALU_PER_MFMA, 0);  // softmax VALU in its shadow
    }
#endif
#ifndef ATT_SKIP_BARRIER
    __syncthreads();
#endif
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

template <int NW, bool HI, bool GLOBAL>
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
  hipLaunchKernelGGL((window_attention_split_kernel<NW, HI, GLOBAL>), dim3(q_tiles * nwr * g.nwc * heads), dim3(NW * 64), 0, st,
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
  const bool whole = (wh == H && ww == W);
  if (!whole && L > MAX_WIN_TOKENS) return CRA5_ERR_ARG;   // attention_f32.hip covers those
#define CRA5_ATT_GO(NWV, HIV, GLV) \
  return launch<NWV, HIV, GLV>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st)
#ifndef ATT_NW_GLOBAL
#define ATT_NW_GLOBAL 12   /* 384 queries share each K/V tile (4 / 8 / 12 waves: 1.59 / 1.56 / 1.52 ms) */
#endif
  if (whole) {
    if (hi_only) CRA5_ATT_GO(ATT_NW_GLOBAL, true, true);
    CRA5_ATT_GO(ATT_NW_GLOBAL, false, true);
  }
  if (L % 192 == 0 && L <= 1152) {
    if (hi_only) CRA5_ATT_GO(6, true, false);
    CRA5_ATT_GO(6, false, false);
  }
  if (hi_only) CRA5_ATT_GO(4, true, false);
  CRA5_ATT_GO(4, false, false);
#undef CRA5_ATT_GO
}
