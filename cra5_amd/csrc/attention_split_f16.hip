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
//                    (p = exp2(s*c - m), c = scale*log2 e folded into one FMA), the running max m
//                    DEFERRED: O / l are rescaled only when a tile max exceeds m by more than 2^8;
//   O^T += V^T . P^T 12 MFMAs: the 16 P values a lane holds ARE its B-operand k-slots (the
//                    reduction index may be visited in any order: slot (t, j) of lane half h is
//                    key (r&3)+8(r>>2)+4h with r = 8t+j), split to hi/lo f16 in registers; V^T
//                    fragments are 2 x ds_read_b64 from a TRANSPOSED [plane][64 d][32 keys] LDS
//                    image (row stride 72 B = 18 dwords: 32 lanes hit 32 distinct even banks).
// 24 MFMAs x 32 cycles = 768 matrix cycles per tile instead of 4096 for the exact-f32 kernel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/cra5_amd.h"
#include "split.h"

CRA5_RANGE_TU(attn)

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

#ifdef CRA5_ATTN_TRACE
// debug build only (tools/attn_trace.py): per work-group (classic) / per unit (PERSIST) wall-clock stamps {start, key loop
// start, key loop end, end}, the key tiles of the unit and the shader cycles between start and end
__device__ unsigned long long g_attn_trace[6 * 8192];
#define CRA5_ATRACE(SLOT, VAL)                                                                          \
  if (threadIdx.x == 0 && atrace_idx < 8192) g_attn_trace[atrace_idx * 6 + (SLOT)] = (VAL);
#else
#define CRA5_ATRACE(SLOT, VAL)
#endif

constexpr int HD = 64;
constexpr int KS = 72;   // K LDS row stride (halves): 144 B
constexpr int VS = 96;   // V LDS row stride (halves): 192 B - see the ds_read_b64_tr_b16 note in the kernel
constexpr int MAX_WIN_TOKENS = 1152;   // windowed (not whole-grid) launches: L <= this (9 KB offset table: three 4-wave
                                        // work-groups of 52.7 KB each fit a CU's 160 KB)

// Balanced schedule of the whole-grid (global) launch (BAL).  A wave owns 32 queries for the whole key loop and the
// matrix pipe is per SIMD, so the unit of work is a (32-query tile, SIMD) pair: 10 368 tokens x 16 heads = 5184
// wave-tiles on 1024 SIMDs = 5.06 each.  The plain launch (27 work-groups of 12 waves per head = 432 on 256 CUs) runs
// two rounds of 3 waves per SIMD = 6 units, the second round on 176 of the 256 CUs.  Here every head owns `groups` =
// CUs / heads work-group slots on ONE XCD (its K / V stay in that L2), and every work-group has NW = 12 waves - three
// per SIMD, the occupancy at which a wave-tile step is cheapest (measured per wave-tile: 208 us with 3 waves per SIMD,
// 249 with 2, 460 with 1):
//   * `n_full` passes of FULL tiles: slot c of pass p runs wave-tiles (p groups + c) NW .. + NW over the whole key
//     loop and stores its output (the same arithmetic as the plain launch: bit-identical);
//   * the remaining R = T - n_full groups NW wave-tiles form ceil(R / NW) tile GROUPS; their group x key-tile steps are
//     laid end to end and cut into `groups` equal PIECES (round 4; round 3 ran them as an 8-wave pass at 2 waves per
//     SIMD plus a key-split 4-wave pass).  Slot c walks piece c - at most two SEGMENTS (group, key range) - and stores
//     un-normalised partials (m, l, O); attention_merge_kernel combines the <= 3 partials of a group in fixed order.
// 10 368 x 16: 192 full tiles + 11 groups x 324 key tiles / 16 slots = 324 + 222.75 steps per CU instead of 324 + 324
// at a lower occupancy.
struct BalArgs {
  int groups;       // work-group slots per head
  int n_full;       // passes of full wave-tiles
  int n_wg_pass;    // heads * groups: work-groups per pass
  int rem_tile0;    // first wave-tile of the key-split part = n_full * groups * NW
  int rem;          // wave-tiles in the key-split part
  int n_grp;        // tile groups in the key-split part = ceil(rem / NW)
  int nkt;          // key tiles of the whole loop = L / 32
  int maxp;         // most pieces overlapping one group
  float *ws;        // partials [head][group][piece slot][NW][32][WS_ROW]
};
// first step of piece c when S steps are cut into G pieces
__host__ __device__ __forceinline__ long bal_cut(long S, int G, int c) { return S * c / G; }
// the piece that holds step s
__host__ __device__ __forceinline__ int bal_piece_of(long S, int G, long s) {
  int c = (int)(s * G / S);
  if (c > G - 1) c = G - 1;
  while (c > 0 && bal_cut(S, G, c) > s) --c;
  while (c < G - 1 && bal_cut(S, G, c + 1) <= s) ++c;
  return c;
}
constexpr int WS_ROW = 68;   // floats per query of a partial: O[64], m, l, 2 pad (16-byte rows)

// PLAIN (reduced-precision mode only, round 5): qkv rows, the pad row and the out_s rows are PLAIN f16 - element n at
// half n of the row (row pitch unchanged: a plain row is the first half of a split row) - as the plain-output qkv GEMM
// writes them and the plain-operand proj GEMM reads them: K / V tiles are staged as 128-byte instead of 256-byte rows.
// PERSIST (round 6): windowed launches as ONE persistent 12-wave work-group per CU that walks a list of UNITS.  A
// (window, head) pair has T = L / 32 wave-tiles of queries (18 for the model's 576-token windows): T / 12 FULL units
// (12 wave-tiles x the whole key loop) and, for the T % 12 = 6 tiles that are left, one SPLIT unit - waves 0..5 run the
// six tiles over the first half of the keys, waves 6..11 the SAME six tiles over the second half (two K / V tiles staged
// per step, one per wave group), and the two un-normalised partials (m, l, O) are merged through LDS in fixed order
// before the store.  Every wave of every unit is busy (the 4-wave form's fifth work-group of a pair was half empty), K / V
// of a pair are staged 1.5 x instead of 5 x, a step runs 3 waves per SIMD behind ONE barrier domain like the whole-grid
// launch, and the per-launch fixed cost (two rounds of prologue + epilogue under contention: 20 of 105 us) is paid once
// per unit on a CU that has nothing else to wait for.  Units are dealt so that the work-groups that got one FULL unit
// more take no SPLIT unit first (576 units on 256 CUs: 36 key steps at most, 30.4 on average).
template <int NW, bool HI, bool GLOBAL, bool BAL = false, bool PLAIN = false, bool PERSIST = false>
__global__ __launch_bounds__(NW * 64, (NW >= 8 ? 1 : (NW == 4 ? 3 : 2))) void window_attention_split_kernel(
    const unsigned short *__restrict__ qkv, long ldq /* halves per row = 2*Kp */,
    const unsigned short *__restrict__ pad_row, float *__restrict__ out, unsigned short *__restrict__ out_s,
    int Kp_out, int C, int heads, WinGeom g, int q_tiles /* PERSIST: number of windows */, float scale, BalArgs bal) {
  static_assert(!BAL || GLOBAL, "the balanced schedule is for whole-grid launches");
  static_assert(!PLAIN || HI, "plain rows carry no lo plane");
  static_assert(!PERSIST || (!GLOBAL && !BAL && NW % 2 == 0), "the persistent unit walk is for windowed launches");
  constexpr int NT = NW * 64;
  constexpr int PPR = PLAIN ? 8 : 16;             // 16-byte pieces per K (or V) row of one head: 64 d x (hi | hi + lo)
  constexpr int PIECES = 32 * PPR;                // 16-byte pieces per K (or V) tile
  constexpr int STG = (PIECES + NT - 1) / NT;
  // PERSIST stages K / V by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) into a ring of NSTG
  // stages per wave group, so that a tile is requested THREE key steps before its scores are due - one step of prefetch
  // (registers) left all 12 waves of the CU waiting for the slowest L2-missing load at every barrier.  A stage is the K
  // image followed by the V image, rows UN-padded (a DMA instruction writes 1 KB linearly: 4 split rows / 8 plain rows),
  // 16-byte pieces XOR-swizzled through the per-lane SOURCE address so that the fragment reads stay conflict-free:
  //   K (ds_read_b128, 16 consecutive rows per cycle):   piece ^ (row & 15)            [plain: piece ^ ((row >> 1) & 7)]
  //   V (ds_read_b64_tr_b16, 4 consecutive rows x 64 B): piece ^ 4 (row & 3)           [plain: piece ^ 4 ((row >> 1) & 1)]
  constexpr int ROWH = PLAIN ? 64 : 128;          // halves per K / V row image
  constexpr int KIMG = 32 * ROWH;                 // halves per K (or V) tile image
  constexpr int STAGE_H = 2 * KIMG;               // K image | V image
  constexpr int NSTG = 4;
  constexpr int NIG = PLAIN ? 8 : 16;             // DMA instructions per stage and wave group (K: first half, V: second)
  constexpr int RPI = PLAIN ? 8 : 4;              // rows per DMA instruction
  constexpr int NPW = (2 * NIG + NW - 1) / NW;    // most DMA instructions a wave issues per stage (SPLIT units: two groups)

  // [K hi][K lo] : 32 x KS ;  [V hi][V lo] : 32 x VS (row-major like K, transposed by the READ)
  constexpr int KPL = 32 * KS + 32;   // K plane stride (halves): +64 B so hi/lo planes hit different bank halves
  constexpr int VPL = 32 * VS;        // V plane stride
  // two K buffers and two V^T buffers: tile j+1's scores are issued to the matrix pipe BEFORE
  // the softmax of tile j, so K runs one tile ahead of V; one barrier per key tile.
  constexpr int KBUF = 2 * KPL, VBUF = 2 * VPL;
  __shared__ __attribute__((aligned(16))) unsigned short lds[PERSIST ? 2 * NSTG * STAGE_H : 2 * KBUF + 2 * VBUF];
  unsigned short *Ks = lds;                       // (PERSIST: wave group g's ring at lds + g * NSTG * STAGE_H)
  unsigned short *Vt = lds + 2 * KBUF;
  // windowed launches: byte offset (from qkv) of every window token's row, pad tokens -> the pad
  // row; built once per block so that the per-tile staging needs no division / multiply.
  constexpr int TAB = GLOBAL ? 1 : MAX_WIN_TOKENS;
  __shared__ long long tab[TAB];

  const int L = g.wh * g.ww;
  // A work-group runs 1 or 2 SEGMENTS = (NW-tile group of queries, key-tile range); only the key-split part of the
  // balanced schedule has two.  seg_part < 0: the segment covers the whole key loop and stores the output.
  int head, win, n_seg = 1;
  int seg_tile0[2] = {0, 0}, seg_nact[2] = {NW, NW}, seg_j0[2] = {0, 0}, seg_j1[2] = {L / 32, L / 32}, seg_part[2] = {-1, -1};
  int seg_grp[2] = {0, 0};
  if (BAL) {
    // blockIdx order IS the schedule: all work-groups of pass 0 are dispatched before any of pass 1.  Inside a pass
    // the work-groups of a head sit on ONE XCD (blockIdx % 8 selects the XCD: its K / V stay in that L2).
    const int pass = blockIdx.x / bal.n_wg_pass, w = blockIdx.x - pass * bal.n_wg_pass;
    int slot;
    if ((heads & 7) == 0 && (bal.n_wg_pass & 7) == 0) {
      const int xcd = w & 7, s_ = w >> 3;
      head = xcd * (heads >> 3) + s_ / bal.groups;
      slot = s_ % bal.groups;
    } else {
      head = w / bal.groups;
      slot = w % bal.groups;
    }
    win = 0;
    if (pass < bal.n_full) {
      seg_tile0[0] = (pass * bal.groups + slot) * NW;
    } else {
      const long S = (long)bal.n_grp * bal.nkt;
      const long s0 = bal_cut(S, bal.groups, slot), s1 = bal_cut(S, bal.groups, slot + 1);
      n_seg = 0;
      for (long s = s0; s < s1;) {
        const int gq = (int)(s / bal.nkt);
        const long e = min(s1, (long)(gq + 1) * bal.nkt);
        if (n_seg < 2) {
          seg_grp[n_seg] = gq;
          seg_tile0[n_seg] = bal.rem_tile0 + gq * NW;
          seg_nact[n_seg] = min(NW, bal.rem - gq * NW);
          seg_j0[n_seg] = (int)(s - (long)gq * bal.nkt);
          seg_j1[n_seg] = (int)(e - (long)gq * bal.nkt);
          seg_part[n_seg] = slot - bal_piece_of(S, bal.groups, (long)gq * bal.nkt);   // index among the group's pieces
          ++n_seg;
        }
        s = e;
      }
    }
  } else if (!PERSIST) {
    const int pid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = pid % q_tiles;
    const int wh_id = pid / q_tiles;
    head = wh_id % heads;
    win = wh_id / heads;
    seg_tile0[0] = qt * NW;
  } else {
    head = 0;
    win = 0;
  }
  int wr = win / g.nwc, wc = win - wr * g.nwc;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  int hoff = head * HD;
  // halves offset of this head's q / k / v slice inside a split row (64 d = 2 chunks = 128 halves)
  long qoff = (PLAIN ? 1L : 2L) * hoff, koff = (PLAIN ? 1L : 2L) * (C + hoff), voff = (PLAIN ? 1L : 2L) * (2 * C + hoff);

  if (!GLOBAL && !PERSIST) {
    const long long pad_delta = reinterpret_cast<const char *>(pad_row) - reinterpret_cast<const char *>(qkv);
    for (int t = tid; t < L; t += NT) {
      const int tok = token_of(g, wr, wc, t);
      tab[t] = (tok >= 0) ? (long long)tok * ldq * 2 : pad_delta;
    }
  }
  constexpr bool IDLE_SKIP = !GLOBAL || BAL;
  bool tab_ready = GLOBAL;
  // PERSIST: the unit walk of this work-group.  T wave-tiles per (window, head) pair = fg FULL units + (rem ? 1 : 0) unit
  // of the rem tiles left (SPLIT when rem = NW / 2).  FULL units are dealt round-robin; the r = nF % G work-groups that got
  // one more of them take no remainder unit before the other G - r have one each.
  const int pu_T = L / 32, pu_fg = pu_T / NW, pu_rem = pu_T - pu_fg * NW;
  const int pu_pairs = PERSIST ? q_tiles * heads : 0;
  const int pu_nF = pu_pairs * pu_fg, pu_nH = pu_rem ? pu_pairs : 0;
  const int pu_G = (int)gridDim.x, pu_r = pu_nF % pu_G;
  const int pu_light = (pu_r && pu_G - pu_r > 0) ? pu_G - pu_r : pu_G;      // work-groups the remainder units rotate over
  int pu_f = (int)blockIdx.x;                                                // next FULL unit of this work-group
  int pu_h = (pu_light == pu_G) ? (int)blockIdx.x : ((int)blockIdx.x >= pu_r ? (int)blockIdx.x - pu_r : pu_nH);
#pragma unroll 1
  for (int seg = 0; PERSIST || seg < n_seg; ++seg) {
  int tile0 = seg_tile0[PERSIST ? 0 : seg], n_active = seg_nact[PERSIST ? 0 : seg], j0 = seg_j0[PERSIST ? 0 : seg], j1 = seg_j1[PERSIST ? 0 : seg];
  bool split = false;         // PERSIST: this unit runs its tiles twice, each wave group over half of the keys
  if (PERSIST) {
    int pair;
    if (pu_f < pu_nF) {
      pair = pu_f / pu_fg;
      tile0 = (pu_f - pair * pu_fg) * NW;
      n_active = NW;
      pu_f += pu_G;
    } else if (pu_h < pu_nH) {
      pair = pu_h;
      tile0 = pu_fg * NW;
      split = (2 * pu_rem == NW) && (pu_T % 2 == 0);
      n_active = pu_rem;
      pu_h += pu_light;
    } else {
      break;
    }
    head = pair % heads;
    win = pair / heads;
    wr = win / g.nwc;
    wc = win - wr * g.nwc;
    hoff = head * HD;
    qoff = (PLAIN ? 1L : 2L) * hoff;
    koff = (PLAIN ? 1L : 2L) * (C + hoff);
    voff = (PLAIN ? 1L : 2L) * (2 * C + hoff);
    j0 = 0;
    j1 = split ? pu_T / 2 : pu_T;
    // the row-offset table of this unit's window (every wave is past the last barrier of the previous unit's key loop,
    // after which nobody reads the table; the barrier below publishes it)
    const long long pad_delta = reinterpret_cast<const char *>(pad_row) - reinterpret_cast<const char *>(qkv);
    for (int t = tid; t < L; t += NT) {
      const int tok = token_of(g, wr, wc, t);
      tab[t] = (tok >= 0) ? (long long)tok * ldq * 2 : pad_delta;
    }
  }
#ifdef CRA5_ATTN_TRACE
  const int atrace_idx = PERSIST ? (int)blockIdx.x * 4 + seg : (GLOBAL ? 8192 : (int)blockIdx.x);
  const unsigned long long atrace_c0 = clock64();
#endif
  CRA5_ATRACE(0, wall_clock64());
  // SPLIT unit: wave group gw = wave / (NW / 2) runs tiles tile0 + wave % (NW / 2) over key tiles [gw * n_tiles, + n_tiles)
  const int gw = (PERSIST && split) ? wave / (NW / 2) : 0;
  const int wq = (PERSIST && split) ? wave - gw * (NW / 2) : wave;
  const int tq = (tile0 + wq) * 32 + l31;
  const int q_tok = (tq < L && wq < n_active) ? token_of(g, wr, wc, tq) : -1;
  const bool wave_active = __any(q_tok >= 0);
  // (also publishes the row-offset table of a windowed launch; between two segments every wave is past the last
  // barrier of the previous key loop, after which nobody reads the K / V buffers any more)
  if (!__syncthreads_or(wave_active ? 1 : 0)) continue;
  tab_ready = true;

  // ---- Q fragments: B operand of S^T = K.Q^T; step s covers d = 16s + 8h + (0..7) ---------
  half8 qh[4], ql[4];
  {
    const unsigned short *qrow = ((q_tok >= 0) ? qkv + (size_t)q_tok * ldq : pad_row) + qoff;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int chunk = s >> 1, pp = 2 * (s & 1) + h;
      if (PLAIN) {
        qh[s] = *reinterpret_cast<const half8 *>(qrow + 16 * s + 8 * h);
        ql[s] = qh[s];   // (never used: HI)
      } else {
        qh[s] = *reinterpret_cast<const half8 *>(qrow + chunk * 64 + pp * 8);
        ql[s] = *reinterpret_cast<const half8 *>(qrow + chunk * 64 + 32 + pp * 8);
      }
    }
  }
  const float cexp = scale * 1.44269504088896340736f;  // scores -> log2 domain
  // Reduced-precision mode: Q is pre-scaled by scale * log2 e ONCE (one more f16 rounding, inside the mode's tolerance)
  // and the first score MFMA of a tile takes -m_run as its C operand, so the accumulator IS the exponent:
  // p = exp2(acc), no v_fma per score (16 VALU fewer per key tile; the fp32-accurate form has no 16 registers to spare
  // for the -m vector - DESIGN.md section 9).  VALU time is wall time on this chip's SIMDs.
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  if (HI) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) qh[s][e] = (_Float16)((float)qh[s][e] * cexp);
  }
  const float cmax = HI ? 1.0f : cexp;   // tile max -> log2 domain (already there in the reduced-precision mode)

  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = HI ? 0.f : -INFINITY, l_run = 0.f;   // (reduced precision: reference 0 until the first tile sets it)

  const int n_tiles = j1 - j0;   // key tiles of this work-group (all L / 32 of them but for the key-split leftover pass)
  uint4 sk0 = make_uint4(0u, 0u, 0u, 0u), sk1 = sk0, sv0 = sk0, sv1 = sk0;
  // K: 16 lanes cover one 256-byte row (coalesced).  V: 32 lanes cover 32 keys at the same
  // 16-byte column, so that the transposed b16 LDS writes of a half-wave land in 32 consecutive
  // halves of ONE V^T row (bank-conflict-free).
  static_assert(STG == 1 || STG == 2, "staging below is written out for one or two 16-byte pieces per thread");
  constexpr bool TWO = STG == 2;
  // thread -> (row, piece) of its two K pieces and (row, two pieces) of V; indexes past the tile
  // (NW = 6: 384 threads x 2 > 512 pieces) are clamped: loaded redundantly, never stored.
  const int krow0 = min(tid / PPR, 31), krow1 = min((tid + NT) / PPR, 31);
  const long kcol = koff + (tid % PPR) * 8;                                 // halves
  const long vcol = voff + (tid % PPR) * 8;
  // whole-grid launches walk three running row pointers (+32 rows per tile); windowed ones look
  // the rows up in the table.
  const unsigned short *kq0 = qkv + (size_t)(krow0 + 32 * j0) * ldq + kcol;
  const unsigned short *kq1 = qkv + (size_t)(krow1 + 32 * j0) * ldq + kcol;
  const unsigned short *vq0 = qkv + (size_t)(krow0 + 32 * j0) * ldq + vcol;
  const unsigned short *vq1 = qkv + (size_t)(krow1 + 32 * j0) * ldq + vcol;
  const long tile_step = 32 * ldq;
#define CRA5_ROW(JJ, ROW) \
  reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(qkv) + tab[(JJ)*32 + (ROW)])
  // (waves whose pieces fall past the 512 of a tile - 4 of 12, or the second piece of 4 of 6 - skip the
  // loads and their address arithmetic altogether: the conditions are wave-uniform)
  const bool stage0 = !PERSIST && ((NT <= PIECES) || (tid < PIECES));   // (PERSIST: no register staging at all)
  const bool stage1 = !PERSIST && TWO && (tid + NT < PIECES);
#define CRA5_K_LOAD(J)                                                                    \
  if (stage0) {                                                                           \
    if (GLOBAL) {                                                                         \
      sk0 = *reinterpret_cast<const uint4 *>(kq0);                                        \
      if (stage1) sk1 = *reinterpret_cast<const uint4 *>(kq1);                            \
      const long st_ = ((J) < n_tiles - 1) ? tile_step : 0; /* past the end: re-read the last tile */ \
      kq0 += st_;                                                                         \
      kq1 += st_;                                                                         \
    } else {                                                                              \
      const int jj_ = min((J), n_tiles - 1);                                              \
      sk0 = *reinterpret_cast<const uint4 *>(CRA5_ROW(jj_, krow0) + kcol);                \
      if (stage1) sk1 = *reinterpret_cast<const uint4 *>(CRA5_ROW(jj_, krow1) + kcol);    \
    }                                                                                     \
  }
#define CRA5_V_LOAD(J)                                                                    \
  if (stage0) {                                                                           \
    if (GLOBAL) {                                                                         \
      sv0 = *reinterpret_cast<const uint4 *>(vq0);                                        \
      if (stage1) sv1 = *reinterpret_cast<const uint4 *>(vq1);                            \
      const long st_ = ((J) < n_tiles - 1) ? tile_step : 0;                               \
      vq0 += st_;                                                                         \
      vq1 += st_;                                                                         \
    } else {                                                                              \
      const int jj_ = min((J), n_tiles - 1);                                              \
      sv0 = *reinterpret_cast<const uint4 *>(CRA5_ROW(jj_, krow0) + vcol);                \
      if (stage1) sv1 = *reinterpret_cast<const uint4 *>(CRA5_ROW(jj_, krow1) + vcol);    \
    }                                                                                     \
  }
  // piece -> (chunk = piece>>3, plane = (piece>>2)&1, d0 = 32*chunk + 8*(piece&3))
#define CRA5_K_STORE1(P, BUF)                                                             \
  {                                                                                       \
    const int idx = tid + (P)*NT;                                                         \
    if (!PERSIST && idx < PIECES) {                                                       \
      const int row = idx / PPR, piece = idx % PPR;                                       \
      const int plane = PLAIN ? 0 : (piece >> 2) & 1, d0 = PLAIN ? 8 * piece : 32 * (piece >> 3) + 8 * (piece & 3); \
      *reinterpret_cast<uint4 *>(Ks + (BUF)*KBUF + plane * KPL + row * KS + d0) = sk##P;  \
    }                                                                                     \
  }
#define CRA5_K_STORE(BUF) { CRA5_K_STORE1(0, BUF) if (TWO) CRA5_K_STORE1(1, BUF) }
#define CRA5_V_STORE1(P, BUF)                                                             \
  {                                                                                       \
    const int idx = tid + (P)*NT;                                                         \
    if (!PERSIST && idx < PIECES) {                                                       \
      const int row = idx / PPR, piece = idx % PPR;                                       \
      const int plane = PLAIN ? 0 : (piece >> 2) & 1, d0 = PLAIN ? 8 * piece : 32 * (piece >> 3) + 8 * (piece & 3); \
      *reinterpret_cast<uint4 *>(Vt + (BUF)*VBUF + plane * VPL + row * VS + d0) = sv##P;  \
    }                                                                                     \
  }
#define CRA5_V_STORE(BUF) { CRA5_V_STORE1(0, BUF) if (TWO) CRA5_V_STORE1(1, BUF) }
  // S^T tile (32 keys x 32 queries) of the K buffer KB: 12 MFMAs on one accumulator (a dependent 32x32x16 MFMA
  // issues back-to-back at full rate: tools/probes/mfma_peak.hip; 2 / 3 independent chains measured the same)
#define CRA5_SCORES(DST, KB)                                                              \
  {                                                                                       \
    f32x16 acc_;                                                                          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc_[r] = 0.f;                         \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                    \
      const half8 kh = PERSIST ? *reinterpret_cast<const half8 *>(ring + (KB)*STAGE_H + koffs[0][st])      \
                               : *reinterpret_cast<const half8 *>(k_base + (KB)*KBUF + 16 * st);            \
      const half8 kl = PERSIST ? *reinterpret_cast<const half8 *>(ring + (KB)*STAGE_H + koffs[PLAIN ? 0 : 1][st]) \
                               : *reinterpret_cast<const half8 *>(k_base + (KB)*KBUF + KPL + 16 * st);      \
      if (!HI) {                                                                          \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[st], acc_, 0, 0, 0);         \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[st], acc_, 0, 0, 0);         \
      }                                                                                   \
      if (HI && st == 0) {                                                                \
        CRA5_MFMA_FROM(acc_, kh, qh[0], negm);                                            \
      } else {                                                                            \
        acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[st], acc_, 0, 0, 0);         \
      }                                                                                   \
    }                                                                                     \
    DST = acc_;                                                                           \
  }
  // D = A.B + C with C a DIFFERENT, persistent register tuple (the -m_run vector): as a builtin hipcc ties D to C and
  // copies the 16 registers in front of every tile.  C is written by an MFMA in the (rare) shift path, which pads the wait
  // states this asm statement does not get from the hazard recogniser (see shift()); the K fragment's s_waitcnt is (operands
  // of the asm statement).
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_MFMA_FROM(D, A, B, C) \
  asm("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(D) : "v"(A), "v"(B), "v"(C))
#else
#define CRA5_MFMA_FROM(D, A, B, C) (D) = (C)
#endif

  const unsigned short *k_base = Ks + l31 * KS + 8 * h;          // + buf*KBUF + plane*KPL + 16*s
  // V^T fragments (A operand: row d = l31, 8 keys per lane) come out of the ROW-MAJOR V image through
  // ds_read_b64_tr_b16: inside each 16-lane group, lane l' receives element (l' & 3) of the 8-byte
  // slots addressed by lanes (l' >> 2) + {0, 4, 8, 12} (probed: tools/probes/tr_probe.hip).  Lane l'
  // therefore ADDRESSES V[kbase + (l' >> 2)][d0 + 4 (l' & 3) ..+3] and RECEIVES V[kbase + 0..3][d0 + l'],
  // d0 = 16 ((lane >> 4) & 1): four consecutive keys of its own d.  Row stride 192 B puts the four
  // key rows x two d-halves of a 32-lane LDS cycle on 8 disjoint 8-bank ranges (conflict-free).
  const unsigned short *v_base = Vt + (4 * h + ((lane & 15) >> 2)) * VS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // + buf*VBUF + plane*VPL + (16*t + 8*a)*VS + 32*dt

  // cross-half max: lanes l and l+32 own the two halves of one query's 32 scores
// (the final v_max_f32 sits INSIDE the asm: `fmaxf` on values hipcc cannot prove canonical - asm outputs, like MFMA
// results - gets a canonicalising `v_max x, x, x` per operand first: 4 of them per key tile in the ISA of round 2)
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_XHALF_MAX(X)                                                                 \
  ({                                                                                      \
    float a_ = (X), b_ = (X), r_;                                                         \
    asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\ts_nop 1\n\tv_max_f32_e32 %2, %0, %1" \
        : "+v"(a_), "+v"(b_), "=v"(r_));                                                  \
    r_;                                                                                   \
  })
// max of two VALU results (never of raw MFMA registers: no hazard wait states are inserted for inline asm)
#define CRA5_MAX2_VALU(A, B)                                                              \
  ({                                                                                      \
    float r_;                                                                             \
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r_) : "v"(A), "v"(B));                          \
    r_;                                                                                   \
  })
#else
#define CRA5_XHALF_MAX(X) (X)
#define CRA5_MAX2_VALU(A, B) fmaxf((A), (B))
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
#define CRA5_TILE_MAX(S)                                                                  \
  ({                                                                                      \
    float m_ = fmaxf(fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3])), fmaxf(fmaxf(S[4], S[5]), fmaxf(S[6], S[7]))); \
    float n_ = fmaxf(fmaxf(fmaxf(S[8], S[9]), fmaxf(S[10], S[11])), fmaxf(fmaxf(S[12], S[13]), fmaxf(S[14], S[15]))); \
    CRA5_XHALF_MAX(fmaxf(m_, n_)) * cmax; /* cexp > 0: max commutes with the scale */      \
  })
#else
// 16 scores -> their max in 8 VALU (v_max3_f32 tree) instead of 15 v_max_f32.  Plain C in the shape hipcc folds
// into v_max3_f32: as inline asm the instruction would read MFMA results without the wait states the hazard
// recogniser inserts for instructions it can see (NaNs).
#define CRA5_MAX3(A, B, C) fmaxf(fmaxf((A), (B)), (C))
#define CRA5_TILE_MAX(S)                                                                  \
  ({                                                                                      \
    const float a_ = CRA5_MAX3(S[0], S[1], S[2]), b_ = CRA5_MAX3(S[3], S[4], S[5]), c_ = CRA5_MAX3(S[6], S[7], S[8]); \
    const float d_ = CRA5_MAX3(S[9], S[10], S[11]), e_ = CRA5_MAX3(S[12], S[13], S[14]);     \
    const float f_ = CRA5_MAX3(a_, b_, c_), g_ = CRA5_MAX3(d_, e_, S[15]);                   \
    CRA5_XHALF_MAX(CRA5_MAX2_VALU(f_, g_)) * cmax; /* cexp > 0: max commutes with the scale */ \
  })
#endif

  // ---- PERSIST: the LDS-DMA ring of this unit ---------------------------------------------------------------------
  // fragment read offsets (halves) inside a stage: K row l31, logical piece P(plane, st) at P ^ swizzle(row); V row
  // 4 h + rr (+ 16 t + 8 a as an immediate), piece Pv(plane, dt) ^ swizzle(row), the lane's 8-byte half of it
  const unsigned short *ring = lds + gw * NSTG * STAGE_H;
  int koffs[2][4], voffs[2][2];
  {
    const int swk = PLAIN ? ((l31 >> 1) & 7) : (l31 & 15);
    const int rr = (lane & 15) >> 2, b_ = (lane >> 4) & 1, c_ = lane & 3;
    const int swv = PLAIN ? 4 * (rr >> 1) : 4 * rr;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int P = PLAIN ? 2 * st + h : 8 * (st >> 1) + 4 * pl + 2 * (st & 1) + h;
        koffs[pl][st] = l31 * ROWH + ((P ^ swk) << 3);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int Pv = PLAIN ? 4 * dt + 2 * b_ + (c_ >> 1) : 8 * dt + 4 * pl + 2 * b_ + (c_ >> 1);
        voffs[pl][dt] = KIMG + (4 * h + rr) * ROWH + ((Pv ^ swv) << 3) + 4 * (c_ & 1);
      }
    }
  }
  // DMA instruction q = wave + NW n of a stage: wave group q / NIG, K (first half) or V, rows RPI i .. of the tile; lane l
  // fetches the logical piece that belongs into physical slot l % PPR of row l / PPR.  dq_*: what stays fixed for the unit.
  const int wave_s = PERSIST ? __builtin_amdgcn_readfirstlane(wave) : 0;
  const int ni_unit = (PERSIST && split) ? 2 * NIG : NIG;
  int dq_row[NPW], dq_tile[NPW], n_w = 0;
  unsigned dq_col[NPW], dq_dst[NPW];
  if (PERSIST) {
    const unsigned lds_b = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned short *)lds);
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
      const int q = wave_s + NW * n;
      const int grp = q / NIG, qq = q - grp * NIG;
      const int isv = qq >= NIG / 2 ? 1 : 0, i_ = qq - isv * (NIG / 2);
      const int r_ = RPI * i_ + lane / PPR, pp = lane % PPR;
      const int sw = isv ? (PLAIN ? 4 * ((r_ >> 1) & 1) : 4 * (r_ & 3)) : (PLAIN ? ((r_ >> 1) & 7) : (r_ & 15));
      dq_row[n] = r_;
      dq_tile[n] = grp * (j1 - j0);
      dq_col[n] = (unsigned)((isv ? voff : koff) * 2) + (unsigned)((pp ^ sw) * 16);
      dq_dst[n] = lds_b + (unsigned)((grp * NSTG * STAGE_H + isv * KIMG) * 2 + i_ * 1024);
      n_w += (q < ni_unit) ? 1 : 0;
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_DMA_ISSUE(T)                                                                 \
  {                                                                                       \
    const int jj_ = min((T), n_tiles - 1);                                                \
    const unsigned slot_ = (unsigned)(((T) & (NSTG - 1)) * STAGE_H * 2);                  \
    _Pragma("unroll") for (int n = 0; n < NPW; ++n) {                                     \
      if (wave_s + NW * n < ni_unit) {                                                    \
        const char *src_ = reinterpret_cast<const char *>(qkv) + tab[(jj_ + dq_tile[n]) * 32 + dq_row[n]] + dq_col[n]; \
        const unsigned dst_ = dq_dst[n] + slot_;                                          \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"   \
                     :: "s"(dst_), "v"(src_) : "memory", "m0");                           \
      }                                                                                   \
    }                                                                                     \
  }
  // own DMA instructions of the stages before the newest one have landed (vmcnt counts in issue order); LATER = 0: all
#define CRA5_DMA_WAIT(LATER)                                                              \
  {                                                                                       \
    const int k_ = (LATER) ? n_w : 0;                                                     \
    if (k_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         \
    else if (k_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                    \
    else if (k_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                    \
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                 \
  }
#else
#define CRA5_DMA_ISSUE(T) (void)dq_row, (void)dq_tile, (void)dq_col, (void)dq_dst
#define CRA5_DMA_WAIT(LATER) (void)n_w
#endif
  static_assert(!PERSIST || NPW <= 3, "CRA5_DMA_WAIT is written out for at most three instructions per wave and stage");

  f32x16 s_cur;
  float mloc;
  if (PERSIST) {
    // prologue: stages 0, 1, 2 requested (every wave is past the barrier that published this unit's row table and has
    // drained its own DMAs / stores of the previous unit); stages 0 and 1 must have landed before the first scores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CRA5_DMA_ISSUE(0);
    if (n_tiles > 1) CRA5_DMA_ISSUE(1);
    if (n_tiles > 2) CRA5_DMA_ISSUE(2);
    CRA5_DMA_WAIT(n_tiles > 2);
    __syncthreads();
    CRA5_SCORES(s_cur, 0);
    mloc = CRA5_TILE_MAX(s_cur);
  } else {
  // prologue: K(0), V(0), K(1) resident; K(2), V(1) in flight; S(0) done
  CRA5_K_LOAD(0);
  CRA5_V_LOAD(0);
  CRA5_K_STORE(0);
  CRA5_V_STORE(0);
  CRA5_K_LOAD(1);
  CRA5_K_STORE(1);
  __syncthreads();
  CRA5_K_LOAD(2);
  CRA5_V_LOAD(1);
  CRA5_SCORES(s_cur, 0);
  mloc = CRA5_TILE_MAX(s_cur);
  // The end of iteration 0 re-fills K buffer 0 with tile 2: every wave must be done reading tile 0 from
  // it (a wave a whole iteration ahead of another is unlikely, not impossible).
  __syncthreads();
  }

  // Waves past the end of the window (q_tok < 0 for all lanes) run the same instruction stream
  // on the pad row and store nothing: one code path, no divergent barriers.
  // One key tile; the loop below is unrolled by two with the score registers swapping roles, so that tile j+1's
  // scores never have to be copied into tile j's registers (8 v_mov_b64 per tile).
  auto key_tile = [&](const int j, f32x16 &s_cur, f32x16 &s_next) __attribute__((always_inline)) {
    const int kb = PERSIST ? ((j + 1) & (NSTG - 1)) : ((j + 1) & 1), vb = PERSIST ? (j & (NSTG - 1)) : (j & 1);
    // PERSIST: stage j + 3 goes into the slot stage j - 1 left at the last barrier (nothing is requested past the unit's
    // last tile: the tail waits for everything instead)
    if (PERSIST && j + 3 < n_tiles) CRA5_DMA_ISSUE(j + 3);
    // (a wave whose 32 queries all lie past the end of the window - half of the fifth 128-query work-group of a
    // 576-token window - only stages and synchronises: its MFMAs would be taken from the other waves of its SIMD)
    if (!IDLE_SKIP || wave_active) {
    // tile j's running max is known before its softmax starts (mloc was reduced in the shadow of
    // the previous tile's PV MFMAs), so the rare O rescale sits at the top and everything below
    // is ONE basic block the scheduler can interleave.
    // DEFERRED running max: m_run moves (and O, l are rescaled) only when some query of the wave saw a tile max more
    // than DEFER_LOG2 above it; until then p = exp2(s c - m_run) may exceed 1, by at most 2^8 - exact in the hi / lo
    // f16 split (its precision is relative; 256 is far inside the f16 range) and in the fp32 accumulators, and O / l
    // carry the same scale, so the result is the softmax whatever reference point the exponentials use.  A new record
    // among the first k keys has probability 1 / k per query: with 32 queries per wave the exact rule (rescale
    // whenever any running max moved) fired on ~38 % of the 324 key tiles of the global launch and on nearly every
    // one of a window's 18; with the threshold it fires on the first tile or two.  (m_run = -inf at the start: the
    // difference is +inf, the branch is taken, alpha = exp2(-inf) = 0 scales the zero accumulators.)
    constexpr float DEFER_LOG2 = 8.0f;
    // Reduced precision: s_cur is RELATIVE to m_run (the -m_run C operand of the score MFMAs).  shift(): move the
    // reference by an f16-representable delta (any point near the maximum will do) - ONE MFMA adds it, exactly, to all
    // 16 registers of a lane (A = e_0 rows: 1 in k-slot 0; B = -delta of the lane's query in k-slot 0).  Updating the
    // C-operand vector by an MFMA keeps it an opaque accumulator tuple for hipcc (written element by element it is
    // re-materialised in front of every tile: 16-31 v_mov per tile).
    auto shift = [&](float delta, const bool rescale, f32x16 *also) __attribute__((always_inline)) {
      delta = (float)(_Float16)fminf(fmaxf(delta, -60000.f), 60000.f);
      if (rescale) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      }
      m_run += delta;
      half8 e0, bd;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        e0[e] = (_Float16)0.0f;
        bd[e] = (_Float16)0.0f;
      }
      e0[0] = (_Float16)(h == 0 ? 1.0f : 0.0f);
      bd[0] = (_Float16)(h == 0 ? -delta : 0.0f);
      s_cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bd, s_cur, 0, 0, 0);
      negm = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bd, negm, 0, 0, 0);
      if (also) *also = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, bd, *also, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
      // negm is the SrcC of the next tile's first score MFMA, which is INLINE ASM (CRA5_MFMA_FROM): no hazard wait states
      // are inserted for it, and an XDL write followed by an overlapped SrcC read of another MFMA needs the writer's passes
      // + 2 (16-pass MFMA: 18).  Spend them here, in the rare shift path, instead of relying on the K-fragment reads that
      // happen to sit in between.
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#endif
    };
    if (HI) {
      // the first tile of a segment sets the reference (either direction) from its max, known from the prologue; later
      // tiles do NOT compute a max at all: see the overflow check behind the exponentials
      if (j == 0) shift(mloc, false, nullptr);
    } else if (!__all(mloc - m_run <= DEFER_LOG2)) {
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    const float m_new = m_run;   // reference point of this tile's exponentials
    // ---- tile j+1's 12 score MFMAs, interleaved with tile j's softmax VALU work
    // (past the last tile the scores of a stale K buffer are computed and discarded.)
    CRA5_SCORES(s_next, kb);
    float psum = 0.f;
    half8 ph[2], pl[2];
#if defined(__HIP_DEVICE_COMPILE__)
    if (HI) {
      // p = exp2(acc); the row sum is taken over the ROUNDED f16 values the PV MFMAs multiply (v_dot2c_f32_f16 with
      // (1, 1): two scores per VALU, fp32 accumulate) - numerator and denominator see the same p.
      // No per-tile max (11 VALU): the exponentials are taken against the standing reference and the row sum tells
      // afterwards whether some p left the safe range (2^13; f16 holds 2^16) - then, and only then, the tile max is
      // computed, O / l / the scores of this and the next tile are shifted and the exponentials redone.  A new record
      // of that size among the first k keys is rare after the first tiles (the fp32-accurate form defers at 2^8).
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      typedef _Float16 half8v __attribute__((ext_vector_type(8)));
      const half2v ones = {(_Float16)1.0f, (_Float16)1.0f};
      auto exps = [&]() __attribute__((always_inline)) {
        psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          half2v hp[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            hp[i] = half2v{(_Float16)__builtin_amdgcn_exp2f(s_cur[8 * t + 2 * i]), (_Float16)__builtin_amdgcn_exp2f(s_cur[8 * t + 2 * i + 1])};
            psum = __builtin_amdgcn_fdot2(hp[i], ones, psum, false);
          }
          ph[t] = half8v{hp[0][0], hp[0][1], hp[1][0], hp[1][1], hp[2][0], hp[2][1], hp[3][0], hp[3][1]};
        }
      };
      exps();
      constexpr float P_SAFE = 8192.0f;
      if (!__all(psum <= P_SAFE)) {          // (a NaN row sum takes the branch too and stays NaN: the range guard's business)
        const float mrel = CRA5_TILE_MAX(s_cur);
        shift(fmaxf(mrel, 0.f), true, &s_next);
        exps();
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) pl[t][e] = (_Float16)0.0f;
    } else
    // p -> (hi, lo) with lo = p - f32(hi) as ONE v_fma_mix_f32 that reads the f16 half in place (no
    // v_cvt_f32_f16 + v_sub per element): 4 VALU per two scores instead of 7.  MFMA and VALU of the waves of a
    // SIMD do not overlap on this chip: every VALU removed here is wall time.  Only the
    // fma_mix is inline asm (hipcc folds fma(x, -1, p) back into a subtraction); the conversions on both sides
    // stay compiler-generated, so the registers the PV MFMAs read are written by instructions whose MFMA
    // wait states the hazard recogniser knows.
    {
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      // (two scores per v_pk_fma_f32 / v_pk_add_f32 instead of scalar fma + add: 16 VALU fewer per tile in the ISA and
      // NOT faster - global 1.39 vs 1.41 ms, windowed 4 % slower: the packed fp32 ops take two issue slots)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(s_cur[r], cexp, -m_new));
        const float p1 = __builtin_amdgcn_exp2f(fmaf(s_cur[r + 1], cexp, -m_new));
        psum += p0;
        psum += p1;
        const half2v h2 = {(_Float16)p0, (_Float16)p1};
        const unsigned hh = __builtin_bit_cast(unsigned, h2);
        float d0, d1;
        // (s_nop: p0 / p1 come out of v_exp_f32, and gfx950 needs a wait state between a transcendental and a
        // VALU that reads its result - inserted by hipcc for its own instructions, not for inline asm: without
        // it one of four window shapes produced garbage)
        asm("s_nop 0\n\tv_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(d0), "=&v"(d1) : "v"(hh), "v"(p0), "v"(p1));
        ph[r >> 3][r & 7] = h2[0];
        ph[r >> 3][(r & 7) + 1] = h2[1];
        pl[r >> 3][r & 7] = (_Float16)d0;
        pl[r >> 3][(r & 7) + 1] = (_Float16)d1;
      }
    }
#else   /* host pass: the same arithmetic without the inline asm (never executed) */
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = HI ? (float)(_Float16)__builtin_amdgcn_exp2f(s_cur[r]) : __builtin_amdgcn_exp2f(fmaf(s_cur[r], cexp, -m_new));
      psum += p;
      const _Float16 hi = (_Float16)p;
      const _Float16 lo = (_Float16)(p - (float)hi);
      ph[r >> 3][r & 7] = hi;
      pl[r >> 3][r & 7] = lo;
    }
#endif
    l_run += psum;
    // ---- O^T += V^T . P^T, with the staging traffic and tile j+1's max in its shadow
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      half8 vh[2], vl[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned short *vp = PERSIST ? ring + vb * STAGE_H + voffs[0][dt] + 16 * t * ROWH : v_base + vb * VBUF + 16 * t * VS + 32 * dt;
        const unsigned short *vq = PERSIST ? ring + vb * STAGE_H + voffs[PLAIN ? 0 : 1][dt] + 16 * t * ROWH : vp + VPL;
        const half4 a0 = CRA5_TR_READ(vp);
        const half4 a1 = CRA5_TR_READ(vp + 8 * (PERSIST ? ROWH : VS));
        const half4 b0 = CRA5_TR_READ(vq);
        const half4 b1 = CRA5_TR_READ(vq + 8 * (PERSIST ? ROWH : VS));
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
    if (!HI) mloc = CRA5_TILE_MAX(s_next);
    }
    // K(j+2) -> the buffer tile j's scores came from (last read before the previous barrier),
    // V(j+1) -> the other V buffer (past the end: stale data into buffers nobody reads);
    // then start fetching K(j+3), V(j+2).
    if (PERSIST) {
      // stage j + 2 (the K tile of the next step's scores) was requested two steps ago: this wave's share of it has
      // landed once at most its share of stage j + 3 - requested at the top of this step, if at all - is outstanding
      CRA5_DMA_WAIT(j + 3 < n_tiles);
    } else {
    CRA5_K_STORE(j & 1);
    CRA5_V_STORE((j + 1) & 1);
    CRA5_K_LOAD(j + 3);
    CRA5_V_LOAD(j + 2);
    }
    __syncthreads();
  };
  CRA5_ATRACE(1, wall_clock64());
  f32x16 s_alt;
  const int n_loop = n_tiles;
  int jt = 0;
  for (; jt + 1 < n_loop; jt += 2) {
    key_tile(jt, s_cur, s_alt);
    key_tile(jt + 1, s_alt, s_cur);
  }
  if (jt < n_loop) key_tile(jt, s_cur, s_alt);
  CRA5_ATRACE(2, wall_clock64());
  CRA5_ATRACE(4, (unsigned long long)n_tiles + ((PERSIST && split) ? 1000 : 0));

  if (PERSIST && split) {
    // merge of the two key halves of a SPLIT unit: wave group 1 parks (m, l) and its 32 accumulator registers in the K / V
    // area (after the key loop's last barrier nobody reads it), 16 registers per round; group 0 combines them with its
    // own - first key half first, fixed association - and stores.  [wq][register][lane] floats: conflict-free.
    float *scr = reinterpret_cast<float *>(lds);
    float ma = 1.f, mb = 1.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (gw == 1) {
        float *dst = scr + (size_t)wq * 18 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = o[t][r];
        if (t == 0) {
          dst[16 * 64] = m_run;
          dst[17 * 64] = l_run;
        }
      }
      __syncthreads();
      if (gw == 0) {
        const float *src = scr + (size_t)wq * 18 * 64 + lane;
        if (t == 0) {
          const float m_b = src[16 * 64], l_b = src[17 * 64];
          const float M = fmaxf(m_run, m_b);
          ma = __builtin_amdgcn_exp2f(m_run - M);
          mb = __builtin_amdgcn_exp2f(m_b - M);
          l_run = fmaf(l_b, mb, l_run * ma);
          m_run = M;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = fmaf(src[r * 64], mb, o[t][r] * ma);
      }
      __syncthreads();
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (BAL && seg_part[PERSIST ? 0 : seg] >= 0) {
    // un-normalised partial of this key range: O, running max (log2 domain), sum - merged by attention_merge_kernel
    if (q_tok >= 0) {
      float *wrow = bal.ws + (((((size_t)head * bal.n_grp + seg_grp[seg]) * bal.maxp + seg_part[seg]) * NW + wave) * 32 + l31) * WS_ROW;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int d = 32 * t + 8 * gq + 4 * h;
          *reinterpret_cast<float4 *>(wrow + d) = make_float4(o[t][4 * gq + 0], o[t][4 * gq + 1], o[t][4 * gq + 2], o[t][4 * gq + 3]);
        }
      if (h == 0) {
        wrow[64] = m_run;
        wrow[65] = l_tot;
      }
    }
    continue;
  }
  // Output.  A query's row is split across the two half-waves: lane (q, h) owns d = 32 t + 8 g + 4 h + (0..3) for
  // g = 0..3.  Stored as they are, that is 16 eight-byte stores per lane for the split layout - and a row-per-lane
  // store tail is ISSUE-bound (MI355X_MICROARCH.md: half the instructions at the same bytes = half the tail).  One
  // v_permlane32_swap per dword turns the groups (g, g + 1) of the two half-waves into 16 contiguous bytes per lane:
  // lanes 0-31 get columns 16 p .. 16 p + 7 of the pair, lanes 32-63 the next eight (8 sixteen-byte stores per lane).
  {
    const bool live = q_tok >= 0 && gw == 0;        // (SPLIT units: wave group 0 holds the merged result)
    const float inv = live ? 1.0f / l_tot : 0.0f;   // (dead lanes take part in the swaps below: keep their values finite)
    float *orow = (out && live) ? out + (size_t)q_tok * C + hoff : nullptr;
    unsigned short *srow = (out_s && live) ? out_s + (size_t)q_tok * 2 * Kp_out : nullptr;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          va[k] = o[t][4 * (2 * pr) + k] * inv;
          vb[k] = o[t][4 * (2 * pr + 1) + k] * inv;
        }
        if (orow) {
          *reinterpret_cast<float4 *>(orow + 32 * t + 16 * pr + 4 * h) = make_float4(va[0], va[1], va[2], va[3]);
          *reinterpret_cast<float4 *>(orow + 32 * t + 16 * pr + 8 + 4 * h) = make_float4(vb[0], vb[1], vb[2], vb[3]);
        }
        if (out_s) {   // (wave-uniform: every lane takes part in the swaps, only the stores are per-lane)
          unsigned ha0, la0, ha1, la1, hb0, lb0, hb1, lb1;
          cra5_split_pair(va[0], va[1], ha0, la0);
          cra5_split_pair(va[2], va[3], ha1, la1);
          cra5_split_pair(vb[0], vb[1], hb0, lb0);
          cra5_split_pair(vb[2], vb[3], hb1, lb1);
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_SWAP32(A, B)                                                  \
  {                                                                        \
    const auto r_ = __builtin_amdgcn_permlane32_swap((A), (B), false, false); \
    (A) = r_[0];                                                           \
    (B) = r_[1];                                                           \
  }
          CRA5_SWAP32(ha0, hb0);
          CRA5_SWAP32(ha1, hb1);
          CRA5_SWAP32(la0, lb0);
          CRA5_SWAP32(la1, lb1);
#undef CRA5_SWAP32
#endif
          if (srow) {
            const int c0 = hoff + 32 * t + 16 * pr + 8 * h;          // first of this lane's 8 columns after the swap
            unsigned short *sp = srow + (PLAIN ? c0 : (c0 >> 5) * 64 + (c0 & 31));
            *reinterpret_cast<uint4 *>(sp) = make_uint4(ha0, ha1, hb0, hb1);
            // (reduced-precision mode: the proj GEMM reads the hi plane only)
            if (!HI) *reinterpret_cast<uint4 *>(sp + 32) = make_uint4(la0, la1, lb0, lb1);
          }
        }
      }
  }
  CRA5_ATRACE(3, wall_clock64());
  CRA5_ATRACE(5, clock64() - atrace_c0);
  }   // segments
  (void)tab_ready;
}

// Merge of the key-split partials: block = (head, group, tile of the group), 256 threads = 32 queries x 8 d-groups of
// 8; piece p of the group contributes exp2(m_p - M) (O_p, l_p), summed in piece order (fixed association: deterministic).
__global__ __launch_bounds__(256) void attention_merge_kernel(const float *__restrict__ ws, float *__restrict__ out,
                                                              unsigned short *__restrict__ out_s, int Kp_out, int C,
                                                              BalArgs bal, int nw, int plain) {
  const int t = blockIdx.x % nw, hg = blockIdx.x / nw;
  const int grp = hg % bal.n_grp, head = hg / bal.n_grp;
  if (grp * nw + t >= bal.rem) return;                        // the last group may be partial
  const long S = (long)bal.n_grp * bal.nkt;
  const int c0 = bal_piece_of(S, bal.groups, (long)grp * bal.nkt);
  const int c1 = bal_piece_of(S, bal.groups, (long)(grp + 1) * bal.nkt - 1);
  const int np = c1 - c0 + 1;
  const int q = threadIdx.x >> 3, d0 = (threadIdx.x & 7) * 8;
  const size_t pstride = (size_t)nw * 32 * WS_ROW;             // floats between two pieces of a group
  const float *base = ws + ((((size_t)head * bal.n_grp + grp) * bal.maxp) * nw + t) * 32 * WS_ROW + (size_t)q * WS_ROW;
  float M = -INFINITY;
  for (int c = 0; c < np; ++c) M = fmaxf(M, base[c * pstride + 64]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lsum = 0.f;
  for (int c = 0; c < np; ++c) {
    const float *r = base + c * pstride;
    const float a = __builtin_amdgcn_exp2f(r[64] - M);
    lsum = fmaf(r[65], a, lsum);
    const float4 v0 = *reinterpret_cast<const float4 *>(r + d0), v1 = *reinterpret_cast<const float4 *>(r + d0 + 4);
    acc[0] = fmaf(v0.x, a, acc[0]);
    acc[1] = fmaf(v0.y, a, acc[1]);
    acc[2] = fmaf(v0.z, a, acc[2]);
    acc[3] = fmaf(v0.w, a, acc[3]);
    acc[4] = fmaf(v1.x, a, acc[4]);
    acc[5] = fmaf(v1.y, a, acc[5]);
    acc[6] = fmaf(v1.z, a, acc[6]);
    acc[7] = fmaf(v1.w, a, acc[7]);
  }
  const float inv = 1.0f / lsum;
  const int tok = (bal.rem_tile0 + grp * nw + t) * 32 + q, col = head * HD + d0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] *= inv;
  if (out) {
    *reinterpret_cast<float4 *>(out + (size_t)tok * C + col) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(out + (size_t)tok * C + col + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  if (out_s) {
    unsigned short *srow = out_s + (size_t)tok * 2 * Kp_out;
    if (plain) {
      cra5_store_plain4(srow, col, acc[0], acc[1], acc[2], acc[3]);
      cra5_store_plain4(srow, col + 4, acc[4], acc[5], acc[6], acc[7]);
    } else {
      cra5_store_split4(srow, col, acc[0], acc[1], acc[2], acc[3]);
      cra5_store_split4(srow, col + 4, acc[4], acc[5], acc[6], acc[7]);
    }
  }
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

constexpr int NW_GLOBAL = 12;   // 384 queries share each K/V tile (4 / 8 / 12 waves per work-group: 1.59 / 1.56 / 1.52 ms)

// geometry of the balanced schedule for L tokens, `heads` heads on this device; false = use the plain launch
bool balanced_plan(int L, int heads, BalArgs &b) {
  const int P = cu_count(), T = L / 32;
  if (P % heads) return false;
  b.groups = P / heads;
  b.n_full = T / (b.groups * NW_GLOBAL);
  b.rem_tile0 = b.n_full * b.groups * NW_GLOBAL;
  b.rem = T - b.rem_tile0;
  b.n_grp = (b.rem + NW_GLOBAL - 1) / NW_GLOBAL;
  b.nkt = T;
  b.n_wg_pass = heads * b.groups;
  b.ws = nullptr;
  b.maxp = 0;
  const long S = (long)b.n_grp * b.nkt;
  for (int g = 0; g < b.n_grp; ++g) {
    const int np = bal_piece_of(S, b.groups, (long)(g + 1) * b.nkt - 1) - bal_piece_of(S, b.groups, (long)g * b.nkt) + 1;
    if (np > b.maxp) b.maxp = np;
  }
  // worth it only when every slot has at least one full pass of wave-tiles, and every piece holds at least one step
  // (S >= groups whenever there is a remainder)
  if (!(b.n_full >= 1 && (b.rem == 0 || (b.n_grp <= b.groups && S >= b.groups)))) return false;
  // The kernel walks a piece as AT MOST TWO (group, key range) segments (its segment builder has two slots): verify it
  // for every piece of THIS plan instead of relying on n_grp <= groups alone - a piece that touched a third group would
  // leave key ranges unvisited and the merge would read partials nobody wrote.
  for (int c = 0; b.rem && c < b.groups; ++c) {
    const long s0 = bal_cut(S, b.groups, c), s1 = bal_cut(S, b.groups, c + 1);
    if (s1 > s0 && (s1 - 1) / b.nkt - s0 / b.nkt + 1 > 2) return false;
  }
  return true;
}

size_t balanced_ws_bytes(const BalArgs &b, int heads) {
  return (size_t)heads * b.n_grp * b.maxp * NW_GLOBAL * 32 * WS_ROW * sizeof(float);
}

template <bool HI, bool PLAIN = false>
int launch_balanced(const unsigned short *qkv, long ldq, const unsigned short *pad_row, float *out, unsigned short *out_s,
                    int Kp_out, int C, int heads, int H, int W, float scale, BalArgs b, float *ws, hipStream_t st) {
  WinGeom g;
  g.H = H;
  g.W = W;
  g.wh = H;
  g.ww = W;
  g.nwc = 1;
  b.ws = ws;
  const int n_pass = b.n_full + (b.rem ? 1 : 0);
  hipLaunchKernelGGL((window_attention_split_kernel<NW_GLOBAL, HI, true, true, PLAIN>), dim3(n_pass * b.n_wg_pass),
                     dim3(NW_GLOBAL * 64), 0, st, qkv, ldq, pad_row, out, out_s, Kp_out, C, heads, g, 0, scale, b);
  int rc = (int)hipGetLastError();
  if (rc || b.rem == 0) return rc;
  hipLaunchKernelGGL(attention_merge_kernel, dim3(heads * b.n_grp * NW_GLOBAL), dim3(256), 0, st, ws, out, out_s, Kp_out,
                     C, b, NW_GLOBAL, PLAIN ? 1 : 0);
  return (int)hipGetLastError();
}

template <int NW, bool HI, bool GLOBAL, bool PLAIN = false>
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
  hipLaunchKernelGGL((window_attention_split_kernel<NW, HI, GLOBAL, false, PLAIN>), dim3(q_tiles * nwr * g.nwc * heads), dim3(NW * 64), 0, st,
                     qkv, ldq, pad_row, out, out_s, Kp_out, C, heads, g, q_tiles, scale, BalArgs{});
  return (int)hipGetLastError();
}

// Windowed launch as persistent 12-wave work-groups walking (window, head) units (PERSIST above): one work-group per CU.
template <bool HI, bool PLAIN = false>
int launch_persist(const unsigned short *qkv, long ldq, const unsigned short *pad_row, float *out, unsigned short *out_s,
                   int Kp_out, int C, int heads, int H, int W, int wh, int ww, float scale, hipStream_t st) {
  WinGeom g;
  g.H = H;
  g.W = W;
  g.wh = wh;
  g.ww = ww;
  const int nwr = (H + wh - 1) / wh;
  g.nwc = (W + ww - 1) / ww;
  const int T = (wh * ww) / 32, pairs = nwr * g.nwc * heads;
  const int units = pairs * (T / NW_GLOBAL + (T % NW_GLOBAL ? 1 : 0));
  const int grid = units < cu_count() ? units : cu_count();
  hipLaunchKernelGGL((window_attention_split_kernel<NW_GLOBAL, HI, false, false, PLAIN, true>), dim3(grid), dim3(NW_GLOBAL * 64),
                     0, st, qkv, ldq, pad_row, out, out_s, Kp_out, C, heads, g, nwr * g.nwc, scale, BalArgs{});
  return (int)hipGetLastError();
}

}  // namespace

static int attention_dispatch(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split, float *out,
                              uint16_t *out_split, int out_kp, int C, int heads, int H, int W, int wh, int ww,
                              float scale, int hi_only, void *workspace, size_t workspace_bytes, void *stream,
                              bool want_balanced) {
  if (!qkv_split || !pad_row_split || (!out && !out_split) || heads <= 0 || C % heads) return CRA5_ERR_ARG;
  if (C / heads != 64 || qkv_kp != 3 * C) return CRA5_ERR_ARG;  // head slices must be chunk-aligned
  if (wh <= 0 || ww <= 0 || H <= 0 || W <= 0 || (wh * ww) % 32) return CRA5_ERR_ARG;
  if (out_split && (out_kp < C || out_kp % 32)) return CRA5_ERR_ARG;
  if (((uintptr_t)qkv_split & 15) || ((uintptr_t)pad_row_split & 15)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int L = wh * ww;
  // hi_only: 0 = fp32-accurate, 1 = reduced precision on split rows, 3 = reduced precision on PLAIN f16 rows (qkv, the
  // pad row and out_split: element n at half n; row pitches unchanged); + CRA5_ATTN_PERSISTENT_UNITS (4): windowed
  // launches as persistent 12-wave units.  The 4-wave work-groups of rounds 2-5 (three per CU) stay the default: the unit
  // walk is built, tested and SLOWER - 117 vs 92 us on the model's 24 x 24 windows (profiles/r06_attn_window_ab.txt): with
  // one work-group per CU nothing runs under a unit's prologue / epilogue (three independent 4-wave work-groups stagger
  // themselves), and a 12-wave barrier domain steps in 2.8 us where the three small ones average 2.2.
  const bool persistent_units = (hi_only & CRA5_ATTN_PERSISTENT_UNITS) != 0;
  hi_only &= ~CRA5_ATTN_PERSISTENT_UNITS;
  if (hi_only != 0 && hi_only != 1 && hi_only != 3) return CRA5_ERR_ARG;
  const bool plain = hi_only == 3;
  const long ldq = 2L * qkv_kp;
  const bool whole = (wh == H && ww == W);
  if (!whole && L > MAX_WIN_TOKENS) return CRA5_ERR_ARG;   // attention_f32.hip covers those
#define CRA5_ATT_GO(NWV, HIV, GLV) \
  return launch<NWV, HIV, GLV>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st)
  if (whole) {
    BalArgs b;
    // (want_balanced: the _ws entry point was used; a plan with no key-split part needs no workspace at all)
    if (want_balanced && balanced_plan(L, heads, b) &&
        (balanced_ws_bytes(b, heads) == 0 ||
         (workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= balanced_ws_bytes(b, heads)))) {
      float *ws = reinterpret_cast<float *>(workspace);
      if (plain)
        return launch_balanced<true, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, scale, b, ws, st);
      if (hi_only)
        return launch_balanced<true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, scale, b, ws, st);
      return launch_balanced<false>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, scale, b, ws, st);
    }
    if (plain) return launch<NW_GLOBAL, true, true, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
    if (hi_only) CRA5_ATT_GO(NW_GLOBAL, true, true);
    CRA5_ATT_GO(NW_GLOBAL, false, true);
  }
  // Windows: 4-wave work-groups (128 queries), three per CU.  The 6-wave form (192 queries: 3 exact blocks per 576-token
  // window, K / V staged three times instead of five) looks better on paper and ran with ONE work-group per CU: its
  // waves land on the SIMDs 2-2-1-1, a second work-group would put four 156-register waves on one SIMD (3 fit), so
  // 864 work-groups took 3.4 rounds instead of 1.7.  Four waves are one per SIMD: three work-groups always fit.
  // (opt-in, round 6: windows of >= 12 wave-tiles as persistent 12-wave units - see windows_persistent())
  if (L / 32 >= NW_GLOBAL && persistent_units) {
    if (plain) return launch_persist<true, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
    if (hi_only) return launch_persist<true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
    return launch_persist<false>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
  }
  if (plain) return launch<4, true, false, true>(qkv_split, ldq, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale, st);
  if (hi_only) CRA5_ATT_GO(4, true, false);
  CRA5_ATT_GO(4, false, false);
#undef CRA5_ATT_GO
}

#ifdef CRA5_ATTN_TRACE
extern "C" int cra5_debug_attn_trace(unsigned long long *host, int n_rows) {
  if (n_rows < 1 || n_rows > 8192) return CRA5_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return CRA5_ERR_ARG;
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_trace), sizeof(unsigned long long) * 6 * n_rows);
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_attn_trace)) == hipSuccess) hipMemset(p, 0, sizeof(unsigned long long) * 6 * 8192);
  return rc;
}
#endif

extern "C" int cra5_window_attention_split(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split,
                                           float *out, uint16_t *out_split, int out_kp, int C, int heads, int H,
                                           int W, int wh, int ww, float scale, int hi_only, void *stream) {
  return attention_dispatch(qkv_split, qkv_kp, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale,
                            hi_only, nullptr, 0, stream, false);
}

extern "C" size_t cra5_attention_workspace_bytes(int n_tokens, int heads) {
  BalArgs b;
  if (n_tokens <= 0 || heads <= 0 || (n_tokens % 32) || !balanced_plan(n_tokens, heads, b)) return 0;
  return balanced_ws_bytes(b, heads);
}

extern "C" int cra5_window_attention_split_ws(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split,
                                              float *out, uint16_t *out_split, int out_kp, int C, int heads, int H,
                                              int W, int wh, int ww, float scale, int hi_only, void *workspace,
                                              size_t workspace_bytes, void *stream) {
  return attention_dispatch(qkv_split, qkv_kp, pad_row_split, out, out_split, out_kp, C, heads, H, W, wh, ww, scale,
                            hi_only, workspace, workspace_bytes, stream, true);
}

extern "C" int cra5_attention_balanced_plan(int n_tokens, int heads, size_t *workspace_bytes) {
  BalArgs b;
  if (workspace_bytes) *workspace_bytes = 0;
  if (n_tokens <= 0 || heads <= 0 || (n_tokens % 32) || !balanced_plan(n_tokens, heads, b)) return 0;
  if (workspace_bytes) *workspace_bytes = balanced_ws_bytes(b, heads);
  return 1;
}
