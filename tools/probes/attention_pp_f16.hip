// NOT PART OF THE BUILD - kept as the record of an experiment (round 2): correct (all attention tests pass with it
// wired into cra5_window_attention_split), bit-identical to window_attention_split_kernel<12, false, true>, and NOT
// faster: 1.40 vs 1.38 ms.  Phase stamps (clock64, tile 100 / 101): every interval lasts 650-940 cycles instead of
// the 384 of its 12 MFMAs - the VALU phases (46 instructions) take 500-800 cycles next to another wave's MFMA
// phase: VALU and MFMA of the waves of one SIMD do not overlap on this chip, only LDS / VMEM work does (which is
// why the same idea pays in the GEMM main loop).  DESIGN.md section 9.
//
// Global (whole-grid) attention on the f16 matrix cores, three-group rotation: the same arithmetic as
// window_attention_split_kernel<12, false, true> (attention_split_f16.hip; reference vit_nlc.py:94-112) - 3 x
// v_mfma_f32_32x32x16_f16 per product on hi / lo split operands, log2-domain online softmax, bit-identical
// results - with the instruction stream of every wave cut into SIX PHASES per 32-key tile and a raw s_barrier at
// every phase boundary:
//
//     A  S^T = K . Q^T        12 MFMAs (K fragments already in registers)
//     B  softmax, first half  running max, rare O rescale, 8 of the 16 exp2 / hi-lo splits
//     C  softmax, second half + the 16 transposed V fragment reads of the tile
//     D  O^T += V^T . P^T     12 MFMAs
//     E  K fragments of the next tile (8 ds_read_b128)        [group 0: + LDS-DMA of K tile j + 3]
//     F  -                                                     [group 1: + LDS-DMA of V tile j + 3]
//
// The block's 12 waves are three groups of four (group = wave / 4: one wave of each group per SIMD), running the
// same loop TWO PHASES APART: in every interval exactly one wave per SIMD is in a matrix phase (A or D), one in a
// VALU phase and one in an LDS / DMA phase.  MFMA, VALU and LDS work of the waves of one SIMD do not overlap well
// when each wave interleaves all three (the free-running kernel keeps the matrix pipes 55 % busy); apart, the pipe
// is handed from wave to wave without a gap.  (The same idea as the GEMM's ping-pong main loop)
//
// K / V tiles live in rings of three LDS buffers, written by LDS-DMA (global_load_lds_dwordx4, inline asm: invisible
// to hipcc's s_waitcnt insertion, drained by explicit vmcnt waits), unpadded 128-byte rows with the bank swizzle on
// the SOURCE side (K: piece ^ ((row >> 1) & 7); V: piece ^ (((row >> 1) & 1) << 2)).  Slots (absolute, group g starts
// tile j at 6 j + 2 g): K(j) is read in [6 j - 2, 6 j + 3], so K(j + 3) is issued into its buffer by group 0 in slot
// 6 j + 4 (its phase E of tile j) and drained (vmcnt) at the end of its phase D of tile j + 2 = slot 6 j + 15, one
// barrier before the first read; V(j) is read in [6 j + 2, 6 j + 6], V(j + 3) is issued by group 1 in slot 6 j + 7 (its
// phase F of tile j) after draining V(j + 2 + ...) - see the loop.  Tiles past the end are clamped to the last one:
// every issue slot issues, so the vmcnt bookkeeping never changes.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/cra5_amd.h"
#include "../../cra5_amd/csrc/split.h"

CRA5_RANGE_TU(attnpp)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
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

constexpr int HD = 64;
constexpr int NW = 12;               // waves per block: 384 queries share every K / V tile
constexpr int ROW = 64;              // LDS row stride (halves): 128 B, unpadded
constexpr int PLANE = 32 * ROW;      // one plane (hi or lo) of a tile: 4 KB
constexpr int TILE = 2 * PLANE;      // [hi][lo]
constexpr int RING = 3;

#define CRA5_PHASE_BARRIER                     \
  {                                            \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("" ::: "memory");             \
  }

__global__ __launch_bounds__(NW * 64, 1) void global_attention_pp_kernel(
    const unsigned short *__restrict__ qkv, long ldq /* halves per row = 2 * Kp */, float *__restrict__ out,
    unsigned short *__restrict__ out_s, int Kp_out, int C, int heads, int L, int q_tiles, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * RING * TILE];
  unsigned short *Ks = lds;
  unsigned short *Vt = lds + RING * TILE;

  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = pid % q_tiles;
  const int head = pid / q_tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wig = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const int hoff = head * HD;
  const long qoff = 2L * hoff, koff = 2L * (C + hoff), voff = 2L * (2 * C + hoff);

  const int tq = (qt * NW + wave) * 32 + l31;
  const int q_tok = (tq < L) ? tq : -1;

  // ---- Q fragments: B operand of S^T = K.Q^T; step s covers d = 16s + 8h + (0..7) ---------
  half8 qh[4], ql[4];
  {
    const unsigned short *qrow = qkv + (size_t)min(tq, L - 1) * ldq + qoff;
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

  // ---- LDS-DMA staging.  A tile is 2 planes x 32 rows x 128 B = 8 one-KB instructions (8 rows x 128 B each),
  // two per wave of the staging group: id = wig + 4 q, plane = id >> 2, rows 8 (id & 3) ..+7.  Lane -> (row =
  // lane >> 3, physical piece = lane & 7); the logical piece it fetches carries the bank swizzle.  Source of
  // (row, plane, piece p): the head's 64-d slice of a split row = 2 chunks [32 hi | 32 lo]: halves offset
  // (p >> 2) * 64 + plane * 32 + (p & 3) * 8.
  const bool stagesV = grp == 1;
  unsigned soff[2];   // byte offset of this lane's 16 bytes relative to the tile's first row, this head's k (v) slice
  unsigned sdst[2];   // LDS byte offset inside a tile image
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int id = wig + 4 * q;
    const int plane = id >> 2, row = (id & 3) * 8 + (lane >> 3), lp8 = lane & 7;
    const int p = stagesV ? (lp8 ^ (((row >> 1) & 1) << 2)) : (lp8 ^ ((row >> 1) & 7));
    soff[q] = (unsigned)(((size_t)row * ldq + (p >> 2) * 64 + plane * 32 + (p & 3) * 8) * 2);
    sdst[q] = (unsigned)(id * 1024);
  }
  const unsigned long long src0 =
      reinterpret_cast<unsigned long long>(qkv + (stagesV ? voff : koff));   // + tile * 32 rows
  const unsigned long long tile_bytes = (unsigned long long)32 * ldq * 2;
  const unsigned lds_ring = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned short *)(stagesV ? Vt : Ks));
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_STAGE_TILE(T, BUF)                                                                   \
  {                                                                                               \
    const unsigned long long base_ = src0 + (unsigned long long)min((T), n_tiles - 1) * tile_bytes; \
    const unsigned dst_ = lds_ring + (unsigned)(BUF) * (TILE * 2);                                \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                 \
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"               \
                   :: "s"(dst_ + sdst[q]), "v"(soff[q]), "s"(base_) : "memory", "m0");            \
  }
#else
#define CRA5_STAGE_TILE(T, BUF) (void)(soff[0] + sdst[0] + src0 + tile_bytes + lds_ring)
#endif

  // K fragment of k16-step st: logical piece 2 st + h of key row l31, at physical piece ^ ((l31 >> 1) & 7)
  const unsigned short *k_base = Ks + l31 * ROW;                 // + buf * TILE + plane * PLANE + kp_off[st]
  int kp_off[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) kp_off[st] = ((2 * st + h) ^ ((l31 >> 1) & 7)) * 8;
  // V^T fragments through ds_read_b64_tr_b16 of the ROW-MAJOR image (attention_split_f16.hip for the lane map);
  // with 128-byte rows the 16-byte piece index is XORed with ((key >> 1) & 1) << 2 = ((lane >> 3) & 1) << 2.
  const int vsk = (lane >> 3) & 1;
  const unsigned short *v_base0 = Vt + (4 * h + ((lane & 15) >> 2)) * ROW + 32 * vsk + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const unsigned short *v_base1 = Vt + (4 * h + ((lane & 15) >> 2)) * ROW + 32 * (vsk ^ 1) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

#define CRA5_XHALF_MAX(X)                                                                 \
  ({                                                                                      \
    float a_ = (X), b_ = (X);                                                             \
    asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(a_), "+v"(b_));              \
    fmaxf(a_, b_);                                                                        \
  })
#define CRA5_MAX3(A, B, C) fmaxf(fmaxf((A), (B)), (C))
#define CRA5_TILE_MAX(S)                                                                  \
  ({                                                                                      \
    const float a_ = CRA5_MAX3(S[0], S[1], S[2]), b_ = CRA5_MAX3(S[3], S[4], S[5]), c_ = CRA5_MAX3(S[6], S[7], S[8]); \
    const float d_ = CRA5_MAX3(S[9], S[10], S[11]), e_ = CRA5_MAX3(S[12], S[13], S[14]);     \
    const float f_ = CRA5_MAX3(a_, b_, c_), g_ = CRA5_MAX3(d_, e_, S[15]);                   \
    CRA5_XHALF_MAX(fmaxf(f_, g_)) * cexp;                                                  \
  })
#define CRA5_K_FRAGS(BUF)                                                                 \
  {                                                                                       \
    _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                    \
      kh[st] = *reinterpret_cast<const half8 *>(k_base + (BUF)*TILE + kp_off[st]);        \
      kl[st] = *reinterpret_cast<const half8 *>(k_base + (BUF)*TILE + PLANE + kp_off[st]); \
    }                                                                                     \
  }
  // exp2 / hi-lo split of scores R0 .. R0 + 7 (one half8 pair): lo = p - f32(hi) as ONE v_fma_mix_f32 reading the f16
  // half in place (only the fma_mix is inline asm: see attention_split_f16.hip for the two hazards)
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_SOFTMAX_HALF(T)                                                              \
  {                                                                                       \
    _Pragma("unroll") for (int r = 8 * (T); r < 8 * (T) + 8; r += 2) {                    \
      const float p0 = __builtin_amdgcn_exp2f(fmaf(s[r], cexp, -m_run));                  \
      const float p1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], cexp, -m_run));              \
      psum += p0;                                                                         \
      psum += p1;                                                                         \
      const half2v h2 = {(_Float16)p0, (_Float16)p1};                                     \
      const unsigned hh = __builtin_bit_cast(unsigned, h2);                               \
      float d0, d1;                                                                       \
      asm("s_nop 0\n\tv_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\t"               \
          "v_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"               \
          : "=&v"(d0), "=&v"(d1) : "v"(hh), "v"(p0), "v"(p1));                            \
      ph[T][r & 7] = h2[0];                                                               \
      ph[T][(r & 7) + 1] = h2[1];                                                         \
      pl[T][r & 7] = (_Float16)d0;                                                        \
      pl[T][(r & 7) + 1] = (_Float16)d1;                                                  \
    }                                                                                     \
  }
#else
#define CRA5_SOFTMAX_HALF(T) (void)psum
#endif

  // ---- prologue: K(0..2) by group 0, V(0..2) by group 1, everybody waits; every wave reads K fragments of tile 0
  if (grp == 0 || grp == 1) {
#pragma unroll
    for (int t = 0; t < RING; ++t) CRA5_STAGE_TILE(t, t);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  half8 kh[4], kl[4];
  CRA5_K_FRAGS(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int i = 0; i < 2 * grp; ++i) CRA5_PHASE_BARRIER;   // group g runs 2 g intervals behind group 0

#ifdef ATT_PP_TRACE   /* debug build: shader-clock stamps of the phase boundaries of tiles 100, 101 (block 0, wave 4 g) */
  __shared__ long long stamps[3][2][13];
#define CRA5_STAMP(PH) if (blockIdx.x == 0 && wig == 0 && lane == 0 && (j == 100 || j == 101)) stamps[grp][j - 100][PH] = clock64();
#define CRA5_STAMPE(PH) CRA5_STAMP(7 + (PH))
#else
#define CRA5_STAMP(PH)
#define CRA5_STAMPE(PH)
#endif
  int buf = 0;                        // j % 3
  for (int j = 0; j < n_tiles; ++j) {
    const int buf1 = (buf == RING - 1) ? 0 : buf + 1;        // (j + 1) % 3
    // ---- A: scores
    CRA5_STAMP(0);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[st], qh[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[st], ql[st], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[st], qh[st], s, 0, 0, 0);
    }
    CRA5_STAMPE(0);
    CRA5_PHASE_BARRIER;
    CRA5_STAMP(1);
    // ---- B: running max, rare rescale, first half of the exponentials
    {
      const float mloc = CRA5_TILE_MAX(s);
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
    }
    float psum = 0.f;
    half8 ph[2], pl[2];
    CRA5_SOFTMAX_HALF(0);
    CRA5_STAMPE(1);
    CRA5_PHASE_BARRIER;
    CRA5_STAMP(2);
    // ---- C: second half of the exponentials, THEN the V fragments (the scores are dead by then: 16 registers
    // fewer at the peak than with the reads issued first)
    CRA5_SOFTMAX_HALF(1);
    l_run += psum;
    __builtin_amdgcn_sched_barrier(0);
    half8 vh[2][2], vl[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const unsigned short *vp = (dt ? v_base1 : v_base0) + buf * TILE + 16 * t * ROW;
        const half4 a0 = CRA5_TR_READ(vp);
        const half4 a1 = CRA5_TR_READ(vp + 8 * ROW);
        const half4 b0 = CRA5_TR_READ(vp + PLANE);
        const half4 b1 = CRA5_TR_READ(vp + PLANE + 8 * ROW);
        vh[t][dt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        vl[t][dt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CRA5_STAMPE(2);
    CRA5_PHASE_BARRIER;
    CRA5_STAMP(3);
    // ---- D: O^T += V^T . P^T
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[t][0], ph[t], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[t][1], ph[t], o[1], 0, 0, 0);
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t][0], pl[t], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t][1], pl[t], o[1], 0, 0, 0);
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t][0], ph[t], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[t][1], ph[t], o[1], 0, 0, 0);
    }
    // group 0: K(j + 1) - issued in phase E of tile j - 2 - must have landed before the next interval reads it;
    // K(j + 2), issued in phase E of tile j - 1, may stay in flight (2 instructions per wave and tile)
    if (grp == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    CRA5_STAMPE(3);
    CRA5_PHASE_BARRIER;
    CRA5_STAMP(4);
    // ---- E: K fragments of tile j + 1; group 0 refills the buffer K(j) just vacated with K(j + 3)
    CRA5_K_FRAGS(buf1);   // (past the last tile: a stale buffer, never used)
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) CRA5_STAGE_TILE(j + 3, buf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CRA5_STAMPE(4);
    CRA5_PHASE_BARRIER;
    CRA5_STAMP(5);
    // ---- F: group 1 (slot 6 j + 7): V(j + 1), issued in phase F of tile j - 2, is first read in slot 6 j + 8 - drain
    // it, leaving V(j + 2) in flight - and refill the buffer V(j) vacated in slot 6 j + 6 with V(j + 3)
    if (grp == 1) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      CRA5_STAGE_TILE(j + 3, buf);
    }
    CRA5_STAMP(6);
    CRA5_PHASE_BARRIER;
    buf = buf1;
  }
#ifdef ATT_PP_TRACE
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0)
    for (int g = 0; g < 3; ++g)
      for (int jj = 0; jj < 2; ++jj)
        printf("grp %d tile %d: slot (work) A %lld (%lld) B %lld (%lld) C %lld (%lld) D %lld (%lld) E %lld (%lld) F - (%lld)\n", g, 100 + jj,
               stamps[g][jj][1] - stamps[g][jj][0], stamps[g][jj][7] - stamps[g][jj][0],
               stamps[g][jj][2] - stamps[g][jj][1], stamps[g][jj][8] - stamps[g][jj][1],
               stamps[g][jj][3] - stamps[g][jj][2], stamps[g][jj][9] - stamps[g][jj][2],
               stamps[g][jj][4] - stamps[g][jj][3], stamps[g][jj][10] - stamps[g][jj][3],
               stamps[g][jj][5] - stamps[g][jj][4], stamps[g][jj][11] - stamps[g][jj][4],
               stamps[g][jj][6] - stamps[g][jj][5]);
#endif
  for (int i = 0; i < 2 * (2 - grp); ++i) CRA5_PHASE_BARRIER;

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

}  // namespace

// Called by cra5_window_attention_split (attention_split_f16.hip) for whole-grid, full-precision launches.
extern "C" __attribute__((visibility("hidden"))) int cra5_internal_global_attention_pp(
    const unsigned short *qkv, long ldq, float *out, unsigned short *out_s, int Kp_out, int C, int heads, int L,
    float scale, hipStream_t st) {
  const int q_tiles = (L + NW * 32 - 1) / (NW * 32);
  hipLaunchKernelGGL(global_attention_pp_kernel, dim3(q_tiles * heads), dim3(NW * 64), 0, st, qkv, ldq, out, out_s,
                     Kp_out, C, heads, L, q_tiles, scale);
  return (int)hipGetLastError();
}
