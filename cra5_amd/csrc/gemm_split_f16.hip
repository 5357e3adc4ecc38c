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
// Kernel structure: tiles of 256x256 / 192x256 (8 waves = 2 per SIMD, 4x2 / 3x2 32x32
// accumulators per wave) or 128x128 / 64x64 (4 waves) chosen per shape; two LDS stages of
// [A rows | W rows], 128 B per row per k-step, filled by LDS-DMA (global_load_lds_dwordx4,
// one cache line per row, XOR swizzle of the 16-byte piece index applied on the source
// side -> conflict-free ds_read_b128 without padding, 256x256 double-buffered = 128 KB);
// main loop of the two big tiles: a PING-PONG between the two waves of each SIMD - one reads the
// fragments of a 16-wide k-half and issues its LDS-DMA while the other issues that half's
// TM*TN*3 MFMAs, a raw s_barrier at every phase boundary (LDS / VMEM work overlaps another
// wave's MFMAs on this chip, VALU does not); the small tiles keep the 2-phase loop (ONE
// barrier per BK = 32 step placed between the step's two 16-wide halves, the next tile's
// first-half fragments prefetched across it); MFMAs issued plane-major (TM*TN independent
// accumulators between dependent MFMAs); XCD-aware tile order in bands of four tile rows;
// epilogue through wave-private LDS (row-contiguous float4 accesses) with fused bias /
// erf-GELU / residual and fp32 and/or split-f16 output - a straight-line body for interior
// tiles (gemm_split_epilogue_fast.inc), the generic one for edges.  Long
// reductions (K > 8192: the patch-embed conv, K = 29 480) are chained through the fp32
// output in chunks of <= 8192 (two-level sum).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/cra5_amd.h"
#include "split.h"

CRA5_RANGE_TU(gemm)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;         // elements per k-step

// gelu(x) = x * Phi(x),  Phi(x) = 0.5 erfc(-x / sqrt 2),  erfc(t) = exp(-t^2) * k P(k), k = 1/(1 + 0.4 t)
// for t >= 0 (degree-7 least-squares fit, |erfc error| <= 8.3e-9 on [0, 6]; evaluated in fp32 the
// gelu error against fp64 is 1.1e-7 RMS / 6.1e-7 max over [-9, 9] - below torch's own fp32
// erf-based gelu, 1.7e-7 / 1.3e-6; tests/test_kernels_gpu.py pins it).  One rcp + one exp2 + 10 FMAs,
// branch-free: the libm erff costs ~3x as much with both of its branches live in a wave.
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

// two values at a time on the packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32): the epilogue has no MFMAs beside it
// and is VALU-bound (the guide's "packed fp32 beside MFMAs is an anti-lever" does not apply here); same
// arithmetic per element as gelu_erf, same operation order -> bit-identical results.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v gelu_erf2(float2v x) {
  const float2v t = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const float2v d = __builtin_elementwise_fma(float2v{0.4f, 0.4f}, t, float2v{1.0f, 1.0f});
  const float2v k = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  float2v p = {0.03080804832279682f, 0.03080804832279682f};
  p = __builtin_elementwise_fma(p, k, float2v{-0.3524225652217865f, -0.3524225652217865f});
  p = __builtin_elementwise_fma(p, k, float2v{1.0205539464950562f, 1.0205539464950562f});
  p = __builtin_elementwise_fma(p, k, float2v{-0.7088391780853271f, -0.7088391780853271f});
  p = __builtin_elementwise_fma(p, k, float2v{0.6733116507530212f, 0.6733116507530212f});
  p = __builtin_elementwise_fma(p, k, float2v{0.0958886444568634f, 0.0958886444568634f});
  p = __builtin_elementwise_fma(p, k, float2v{0.2406993806362152f, 0.2406993806362152f});
  const float2v a = -(t * t) * 1.4426950408889634f;
  const float2v e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const float2v half_erfc = 0.5f * p * k * e;
  const float2v phi = {(x[0] >= 0.f) ? 1.0f - half_erfc[0] : half_erfc[0], (x[1] >= 0.f) ? 1.0f - half_erfc[1] : half_erfc[1]};
  return x * phi;
}

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb / 8, r = nb % 8;
  const int xcd = bid % 8, within = bid / 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

#ifdef CRA5_GEMM_TRACE
// debug build only (tools/gemm_trace.py): per-block timestamps {entry, main loop start, main loop
// end, exit} from the 100 MHz wall clock, + the hardware id register.
__device__ unsigned long long g_gemm_trace[5 * 8192];
#define CRA5_TRACE(slot)                                                        \
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * 5 + (slot)] = wall_clock64();
#else
#define CRA5_TRACE(slot)
#endif

// Geometry of the fused un-embed epilogue (UE = true, gemm_split_epilogue_unembed.inc): C is then the reconstruction
// x[C][H][W] itself.
struct UnembedArgs {
  float *side;                 // [C][Hp][2][W]: the ky = 0 / ky = 10 rows, un-normalised (cra5_unembed_fixup adds the pairs)
  const float *mean, *stdv;    // per channel, or both null (normalised output)
  int H, W, Hp, Wp;
};

template <int WM, int WN, int TM, int TN, bool LONGK, int STAGES = 2, int NPROD = 3, bool UE = false>
__global__ __launch_bounds__(WM *WN * 64, ((STAGES == 1 || (WM * WN == 4 && TM == 3)) ? 2 : 1)) void gemm_nt_split_kernel(
    const unsigned short *__restrict__ A, long lda, const unsigned short *__restrict__ W, long ldw, float *C,
    int ldc, unsigned short *Cs, long ldcs, const float *__restrict__ bias, const float *res, int ldr, int M,
    int N, int Kp, float wscale_inv, int flags, int tiles_n, UnembedArgs ue) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NT = WM * WN * 64;
  constexpr int ROWS_PER_PASS = NT / 8;  // 8 x 16 B per row per k-step (hi 64 B | lo 64 B)
  constexpr int A_P = BM / ROWS_PER_PASS;
  constexpr int B_P = BN / ROWS_PER_PASS;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile/threads mismatch");
  constexpr int STAGE = (BM + BN) * 64;  // halves per pipeline stage: 128 B per row

  // two stages of [A rows][W rows], 128 B per row = [32 hi | 32 lo] of one k-step, XOR-swizzled in
  // 16-byte pieces (see the staging comment below); 2 x (BM + BN) x 128 B, no padding.
  __shared__ __attribute__((aligned(16))) unsigned short lds[STAGES * STAGE];

  CRA5_TRACE(0);
#ifdef CRA5_GEMM_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * 5 + 4] = clock64();
#endif
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x;
  // the wave index as a SCALAR: everything derived from it - the LDS-DMA destinations above all - stays in SGPRs
  // (m0 = s_add instead of v_add_u32 + v_readfirstlane_b32 + s_mov per 1-KB DMA instruction; 42 -> 28 instructions per
  // k-step of staging.  Measured, interleaved A/B: qkv 184 -> 175 us, fc1 279 -> 270, un-embed 1776 -> 1670)
  const int lane = tid & 63, wave = (NPROD >= 2) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int wave_dma = wave;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  // NPROD == 2 (reduced-precision mode, wide form): a k-step is 64 k-values - the hi halves of TWO chunks - staged into
  // the same 128-byte LDS rows the fp32-accurate mode uses ([32 hi of chunk 2s | 32 hi of chunk 2s + 1] instead of
  // [32 hi | 32 lo]): same LDS-DMA instruction count, same fragment reads and the same ping-pong loop per k-step, two
  // MFMAs per accumulator and 16-wide k-block instead of three, and K advances twice as fast.
  constexpr int KSTEP = (NPROD == 2) ? 2 * BK : BK;
  const int nk_all = Kp / KSTEP;
  // ... and (round 5) an operand of that form may be a PLAIN f16 matrix (CRA5_GEMM_A_PLAIN / _W_PLAIN): a row is Kp
  // contiguous halves, so a 64-wide k-step of a row is ONE full 128-byte line instead of the hi halves of two split
  // chunks (two half-lines).  An LDS-DMA instruction then asks for 8 lines instead of 16 half-lines, which is what the
  // wide form's read phase pays for: -13..-18 % per launch (profiles/r05_f16_plain_layout_experiment.txt).  Same LDS
  // image, same fragment reads, same arithmetic - bit-identical results.  CRA5_GEMM_OUT_PLAIN: C_split row = N
  // contiguous halves (the layout the next reduced-precision GEMM / the attention kernel reads).
  const bool a_plain = (NPROD == 2) && (flags & CRA5_GEMM_A_PLAIN);
  const bool w_plain = (NPROD == 2) && (flags & CRA5_GEMM_W_PLAIN);
  const bool o_plain = (NPROD != 3) && (flags & CRA5_GEMM_OUT_PLAIN);

  const int tile = pid, ka = 0, kb = nk_all;
  // Tile order: the 32 work-groups an XCD runs at a time own 32 CONSECUTIVE tile numbers (xcd_remap), so tiles are
  // numbered in bands of GEMM_GROUP_M tile rows, column-major inside a band: 32 consecutive tiles = 4 A panels x 8 W
  // panels through that XCD's L2 per round (12 panel reads) instead of 2 x 16 (18) in row-major order.
  constexpr int GEMM_GROUP_M = 4;
  int tm, tn;
  if (GEMM_GROUP_M > 1) {
    const int tiles_m = (M + BM - 1) / BM;
    const int band = tile / (GEMM_GROUP_M * tiles_n), r = tile - band * (GEMM_GROUP_M * tiles_n);
    const int rows = min(tiles_m - band * GEMM_GROUP_M, GEMM_GROUP_M);
    tn = r / rows;
    tm = band * GEMM_GROUP_M + (r - tn * rows);
  } else {
    tm = tile / tiles_n;
    tn = tile % tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // Staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves 1 KB straight into LDS -
  // no staging VGPRs, no ds_write pass.  The destination is wave-uniform base + lane*16, i.e. LINEAR
  // (row = lane/8, physical piece = lane%8), so the XOR swizzle is applied to the per-lane SOURCE
  // address.  A stage is [A rows][W rows] = (BM + BN)/8 one-KB groups, dealt round-robin to the waves.
  // Rows past M / N are clamped to the last row (duplicates, never stored).
  // NPROD == 1 (reduced-precision mode): only the hi plane is staged - 16 rows x 64 B (the first half of
  // each line) per instruction into 64-byte LDS rows, piece p of row r at p ^ ((r >> 2) & 3): half the
  // LDS-DMA traffic of the fp32-accurate mode, which is what bounds this mode (1/3 of the MFMAs).
  constexpr int ROWB = (NPROD >= 2) ? 64 : 32;                   // halves per LDS row
  constexpr int GROUPS = (BM + BN) * ROWB / 512;                 // 1-KB groups per stage
  constexpr int NWAVE = WM * WN;
  constexpr int IPW = (GROUPS + NWAVE - 1) / NWAVE;              // LDS-DMA instructions per wave per k-step
  static_assert(NPROD == 1 || GROUPS % NWAVE == 0, "groups must divide evenly over the waves");
  const unsigned short *src[IPW];
  // Hand-written LDS-DMA with the saddr + voffset encoding: a uniform 64-bit cursor per operand (advanced by an
  // s_add per k-step) + a fixed 32-bit byte offset per lane and instruction - no address VALU in the k-loop (the
  // builtin form spends a v_lshl_add_u64 per instruction per k-step, and hipcc does not pick this encoding for it).
  // Measured, interleaved A/B: qkv 182 -> 176 us, un-embed 1762 -> 1736, proj 75.6 -> 72.1, fc2 251 -> 245,
  // patch-embed chunk 459 -> 438.  The loads are invisible to hipcc's s_waitcnt insertion: the prologue and
  // CRA5_K_BARRIER wait with an explicit s_waitcnt vmcnt(0) (exactly the DMA of the tile the barrier publishes is
  // outstanding there).  The per-lane offsets are relative to the tile's first row (< 256 rows x the row pitch: the
  // launcher refuses pitches of 2^21 k-columns (2^22 halves) and more, far above anything the path has).
  constexpr bool ASM_DMA = ((TM == 4 || TM == 3) && NPROD >= 2 && WM == 2 && WN == 4 && !LONGK);
  static_assert(!ASM_DMA || BM % (WM * WN * 8) == 0, "A / W staging instructions must not straddle");
  // the generic staging path below advances a source by 2 * KSTEP halves per k-step, which is right for split rows only:
  // plain-row operands (NPROD == 2) exist with the hand-written DMA's kstride_a / kstride_w alone
  static_assert(NPROD != 2 || ASM_DMA, "the wide reduced-precision form needs the ASM_DMA staging (plain-row k-stride)");
  unsigned soff[IPW];
  unsigned long long curA = 0, curW = 0;
  // NPROD == 3: one instruction = 8 rows x 128 B: a row's [32 hi | 32 lo] chunk is one cache line, requested
  // once (16 rows x 64 B of one plane per instruction asked for every line twice: -3..6 % on the 192x256
  // tiles).  LDS rows are 128 B = 8 pieces [hi 0-3 | lo 4-7]; piece p of row r lives at physical piece
  // p ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-byte slots.
  {
    constexpr int RPG = 512 / ROWB;                              // rows per group: 8 | 16
    constexpr int LPR = 64 / RPG;                                // lanes per row: 8 | 4
    const int lrow = lane / LPR;
#pragma unroll
    for (int q = 0; q < IPW; ++q) {
      const int gid = min(wave + q * NWAVE, GROUPS - 1);         // (NPROD == 1, 192x256: 28 groups on 8 waves)
      const int grow = gid * RPG;
      const bool isA = grow < BM;
      const int row_ = (isA ? grow : grow - BM) + lrow;
      const int lpiece = (NPROD >= 2) ? ((lane & 7) ^ ((row_ >> 1) & 7)) : ((lane & 3) ^ ((row_ >> 2) & 3));
      // halves offset of the logical piece inside the row: contiguous [hi | lo] of one chunk, or (NPROD == 2) the hi
      // half of chunk 0 (pieces 0-3) and of chunk 1 (pieces 4-7)
      const int poffs = (NPROD == 2 && !(isA ? a_plain : w_plain)) ? ((lpiece >> 2) * 64 + (lpiece & 3) * 8) : lpiece * 8;
      if (isA)
        src[q] = A + (size_t)min(m0 + row_, M - 1) * lda + poffs + (size_t)ka * 64;
      else
        src[q] = W + (size_t)min(n0 + row_, N - 1) * ldw + poffs + (size_t)ka * 64;
      if (ASM_DMA)
        soff[q] = (unsigned)((((size_t)(isA ? (size_t)(min(m0 + row_, M - 1) - m0) * lda : (size_t)(min(n0 + row_, N - 1) - n0) * ldw)) + poffs) * 2);
    }
  }
  if (ASM_DMA) {
    curA = reinterpret_cast<unsigned long long>(A + (size_t)m0 * lda) + (unsigned long long)ka * 128;
    curW = reinterpret_cast<unsigned long long>(W + (size_t)n0 * ldw) + (unsigned long long)ka * 128;
  }
  // (the builtin only exists in the device pass; the host pass just needs the launch stub)
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_GLDS16(SRC, DST) __builtin_amdgcn_global_load_lds(SRC, DST, 16, 0, 0)
#else
#define CRA5_GLDS16(SRC, DST) (void)(SRC)
#endif
#define CRA5_DMA_KSTRIDE (2 * KSTEP * 2)   /* bytes along a row per k-step: 128 B per 32-wide chunk (hi + lo) */
  const unsigned kstride_a = a_plain ? 128u : (unsigned)CRA5_DMA_KSTRIDE, kstride_w = w_plain ? 128u : (unsigned)CRA5_DMA_KSTRIDE;
#if defined(__HIP_DEVICE_COMPILE__)
#define CRA5_STAGE_LOAD(BUF)                                                                   \
  {                                                                                            \
    if (ASM_DMA) {                                                                             \
      /* BM = BN = 256, 8 waves: instructions q < 4 stage A rows, q >= 4 stage W rows */       \
      const unsigned ldsb_ = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned short *)(lds + (BUF)*STAGE)); \
      _Pragma("unroll") for (int q = 0; q < IPW; ++q) {                                        \
        const unsigned dst_ = ldsb_ + (wave_dma + q * NWAVE) * 1024;                           \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"          \
                     :: "s"(dst_), "v"(soff[q]), "s"((q * NWAVE * 8 < BM) ? curA : curW) : "memory", "m0"); \
      }                                                                                        \
      curA += kstride_a;                                                                       \
      curW += kstride_w;                                                                       \
    } else {                                                                                   \
      _Pragma("unroll") for (int q = 0; q < IPW; ++q) {                                        \
        if (GROUPS % NWAVE == 0 || wave + q * NWAVE < GROUPS)                                  \
          CRA5_GLDS16(src[q], lds + (BUF)*STAGE + (wave_dma + q * NWAVE) * 512);               \
        src[q] += 2 * KSTEP;                                                                   \
      }                                                                                        \
    }                                                                                          \
  }
#else
#define CRA5_STAGE_LOAD(BUF)                                                                   \
  {                                                                                            \
    _Pragma("unroll") for (int q = 0; q < IPW; ++q) {                                          \
      if (GROUPS % NWAVE == 0 || wave + q * NWAVE < GROUPS)                                    \
        CRA5_GLDS16(src[q], lds + (BUF)*STAGE + (wave_dma + q * NWAVE) * 512);                 \
      src[q] += 2 * KSTEP;  /* next k-step: 128 B (one chunk) or 256 B further along the row */ \
    }                                                                                          \
  }
#endif

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

  // fragment read offsets (halves) inside a stage: row * 64 + (piece ^ sw) * 8, piece = 2*kk + h (hi)
  // or 4 + 2*kk + h (lo); sw = (row >> 1) & 7 = (l31 >> 1) & 7 (sub-tile bases are multiples of 32 rows)
  // (NPROD == 1: 64-byte rows, 4 pieces, sw = (row >> 2) & 3)
  const int sw = (NPROD >= 2) ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
  const int a_row = (wm * TM * 32 + l31) * ROWB;
  const int b_row = BM * ROWB + (wn * TN * 32 + l31) * ROWB;
  int poff[2], poff_lo[2];
  poff[0] = ((0 + h) ^ sw) << 3;
  poff[1] = ((2 + h) ^ sw) << 3;
  poff_lo[0] = ((4 + h) ^ sw) << 3;
  poff_lo[1] = ((6 + h) ^ sw) << 3;
  constexpr int A_LO = 0, B_LO = 0, SUB = 32 * ROWB;

  // fragments of one 16-wide k-half (KK = 0 | 1) of stage ST
#define CRA5_FRAG_READ(AH, AL, BH, BL, ST, KK)                                                 \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                           \
      AH[i] = *reinterpret_cast<const half8 *>((ST) + a_row + i * SUB + poff[KK]);             \
      if (NPROD >= 2) AL[i] = *reinterpret_cast<const half8 *>((ST) + a_row + A_LO + i * SUB + poff_lo[KK]); \
    }                                                                                          \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                           \
      BH[j] = *reinterpret_cast<const half8 *>((ST) + b_row + j * SUB + poff[KK]);             \
      if (NPROD >= 2) BL[j] = *reinterpret_cast<const half8 *>((ST) + b_row + B_LO + j * SUB + poff_lo[KK]); \
    }                                                                                          \
  }
  // small terms first, plane-major: TM*TN independent accumulators between dependent MFMAs.
  // NPROD == 1 is the reduced-precision mode (BASELINE.json configs[4]): hi.hi only = plain f16.
#define CRA5_MFMA_GROUP(AH, AL, BH, BL)                                                        \
  {                                                                                            \
    if (NPROD == 2) {   /* "lo" registers = the hi halves of the step's second chunk: k-block KK + 2 */ \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BL[j], acc[i][j], 0, 0, 0); \
    }                                                                                          \
    if (NPROD == 3) {                                                                          \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL[i], BH[j], acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BL[j], acc[i][j], 0, 0, 0); \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH[i], BH[j], acc[i][j], 0, 0, 0);  \
  }

  // Main loop, ONE barrier per k-step, placed between the two 16-wide halves so that nothing
  // waits right after it:
  //     read F1 = second-half fragments of tile kt | MFMA(F0)
  //     barrier   (own LDS-DMA of tile kt+1 drained first: it was issued a full step ago)
  //     read F0 = first-half fragments of tile kt+1, start the LDS-DMA of tile kt+2 into the
  //     stage tile kt just vacated | MFMA(F1)
  // Measured (tools/gemm_trace.py variants): with the barrier at the end of the step the matrix
  // pipe idled through barrier skew + the first ds_reads of the new stage, 20 of 78 us per tile.
  const int nk = kb - ka;
  half8 f0ah[TM], f0al[TM], f0bh[TN], f0bl[TN], f1ah[TM], f1al[TM], f1bh[TN], f1bl[TN];
  constexpr bool PP = ASM_DMA;
  // Ping-pong main loop (PP; the 256 x 256 and 192 x 256 instantiations): the two waves of a SIMD (wave w and w + 4:
  // wm = 0 | 1) run the same four phases per k-step - read the fragments of a 16-wide k-half, MFMA it, read the
  // other half, MFMA it - ONE PHASE APART, a raw s_barrier at every phase boundary: while one wave of the SIMD feeds
  // the matrix pipe (TM x TN x 3 back-to-back MFMAs) the other does its ds_reads and its LDS-DMA issue and waits at
  // the barrier.  MFMA and the LDS / VMEM work of the partner wave do not overlap on a SIMD on this chip (DESIGN.md
  // section 9), and a wave that issues its DMA burst next to its own MFMAs stalls them (the asymmetric variant -
  // wm = 1 issuing at the top of an MFMA phase - was 2-5 % slower than the 2-phase loop below; this one is 3-10 %
  // faster: qkv 169 -> 164 us, fc1 263 -> 250, fc2 237 -> 216, patch-embed chunk 436 -> 391, un-embed 1690 -> 1623).
  // One fragment register set instead of two.
  // Intervals t = 4 kt + {0, 1, 2, 3} (wm = 0) and one later (wm = 1).  The DMA of tile kt + 1 goes into the stage
  // tile kt - 1 vacated (its last read: wm = 1, interval 4 kt - 1, retired by the lgkmcnt(0) before that barrier) and
  // is issued by every wave in its first read phase of tile kt (intervals 4 kt and 4 kt + 1); each wave drains its
  // own DMA (vmcnt(0)) before the barrier that ends interval 4 kt + 3 - wm = 0 behind its second MFMA phase, wm = 1 in
  // its second read phase - and the first read of tile kt + 1 is in interval 4 kt + 4.
#define CRA5_PP_BARRIER                        \
  {                                            \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("" ::: "memory");             \
  }
#define CRA5_PP_READ(ST, KK) CRA5_FRAG_READ(f0ah, f0al, f0bh, f0bl, ST, KK)
#define CRA5_PP_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  if (PP) {
    // A wave group whose rows all lie past M (the lower half of the 41st tile row of the model's 10 368-token launches:
    // 10 368 = 40.5 x 256) keeps staging and synchronising but issues no MFMAs: its accumulators are never stored, and
    // under the chip's power limit an MFMA that computes nothing is not free.
    const bool rows_live = m0 + wm * TM * 32 < M;
    CRA5_STAGE_LOAD(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CRA5_TRACE(1);
    // (NPROD == 2, the reduced-precision wide form, runs the same loop: its 16-MFMA phases (512 cycles) no longer cover the
    // partner's read phase (~870 cycles: 3470 per k-step for 2048 of MFMAs, tools/gemm_trace.py --hi).  Two re-arrangements
    // were measured and lost - two phases per k-step with double fragment registers (qkv tile main loop 35.4 vs 29.2 us:
    // the eight LDS-DMA issues per wave and k-step dominate a read phase, whoever issues them) and the DMA issued at the top
    // of the first MFMA phase (31.7 us): DESIGN.md section 9.)
    {
    if (wm == 1) CRA5_PP_BARRIER;   // the second group runs one interval behind
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned short *st = lds + (kt & 1) * STAGE;
      if (rows_live) CRA5_PP_READ(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) CRA5_STAGE_LOAD((kt + 1) & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CRA5_PP_BARRIER;
      if (rows_live) CRA5_MFMA_GROUP(f0ah, f0al, f0bh, f0bl);
      CRA5_PP_BARRIER;
      if (rows_live) CRA5_PP_READ(st, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wm == 1) CRA5_PP_DRAIN;
      CRA5_PP_BARRIER;
      if (rows_live) CRA5_MFMA_GROUP(f0ah, f0al, f0bh, f0bl);
      if (wm == 0) CRA5_PP_DRAIN;
      CRA5_PP_BARRIER;
    }
    }
    if (wm == 0) CRA5_PP_BARRIER;
  } else {
  CRA5_STAGE_LOAD(0);
  __syncthreads();   // (hipcc drains vmcnt before the barrier: tile 0 has landed for everyone)
  CRA5_FRAG_READ(f0ah, f0al, f0bh, f0bl, lds, 0);
  if (nk > 1) CRA5_STAGE_LOAD(1);
  CRA5_TRACE(1);

  // One k-step (an unroll by two with literal stage indexes measured 1.5-3 % slower: DESIGN.md section 9).
#define CRA5_K_STEP(KT, CUR)                                                                     \
  {                                                                                              \
    CRA5_FRAG_READ(f1ah, f1al, f1bh, f1bl, lds + (CUR)*STAGE, 1);                                \
    CRA5_MFMA_GROUP(f0ah, f0al, f0bh, f0bl);                                                     \
    CRA5_K_BARRIER;   /* tile kt+1 visible to everyone; everyone is done reading tile kt's stage */ \
    if ((KT) + 1 < nk) CRA5_FRAG_READ(f0ah, f0al, f0bh, f0bl, lds + ((CUR) ^ 1) * STAGE, 0);     \
    /* (dealing the DMA instructions out between the MFMAs instead of issuing them here in a burst  \
       measured the same: 76-78 us per tile either way) */                                       \
    if ((KT) + 2 < nk) CRA5_STAGE_LOAD(CUR);                                                     \
    CRA5_MFMA_GROUP(f1ah, f1al, f1bh, f1bl);                                                     \
    if (LONGK && (((KT) & 15) == 15)) {                                                          \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                       \
            master[i][j][r] += acc[i][j][r];                                                     \
            acc[i][j][r] = 0.f;                                                                  \
          }                                                                                      \
    }                                                                                            \
  }
#define CRA5_K_BARRIER __syncthreads()
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    CRA5_K_STEP(kt, cur);
  }
  }   // !PP
  __syncthreads();   // the epilogue reuses the stages as scratch

  CRA5_TRACE(2);
  const bool has_bias = flags & CRA5_EPI_BIAS;
  const bool do_gelu = flags & CRA5_EPI_GELU;
  const bool has_res = flags & CRA5_EPI_RES;

  // Epilogue through wave-private LDS (the pipeline stages are free after the last barrier):
  // a wave parks one 32 x (TN*32) row block of accumulators, reads it back row-contiguous, 4
  // columns per lane, and applies scale / bias / GELU / residual on float4s: 16-byte residual
  // loads and fp32 stores, 8-byte packed hi / lo stores of the split layout - 8 memory
  // instructions per 32 rows instead of 32-64 dword ones.
  constexpr int COLS = TN * 32;
  constexpr int LDW = COLS + 4;          // floats; +4 keeps rows 16-byte aligned and shifts banks
  constexpr int LPR = COLS / 4;          // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per wave-instruction
  static_assert((size_t)WM * WN * 32 * LDW * 4 <= sizeof(lds), "epilogue staging does not fit the stages");
  float *stg = reinterpret_cast<float *>(lds) + wave * 32 * LDW;
  const int er = lane / LPR, ec = (lane % LPR) * 4;
  const int nw = n0 + wn * COLS + ec;    // first of this lane's 4 columns
  const bool vecC = C && ((ldc & 3) == 0) && ((reinterpret_cast<size_t>(C) & 15) == 0);
  const bool vecR = has_res && ((ldr & 3) == 0) && ((reinterpret_cast<size_t>(res) & 15) == 0);
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (nw + c < N) bv[c] = bias[nw + c];
  }
  if constexpr (UE) {
#include "gemm_split_epilogue_unembed.inc"
  } else {
    // interior tiles of the four epilogue shapes the model uses take the straight-line body; everything else (edge
    // tiles, unaligned outputs, both outputs at once, long-K master accumulators) the generic one
    const bool interior = !LONGK && m0 + BM <= M && n0 + BN <= N;
    const bool c_only = C && !Cs && vecC && !do_gelu;
    const bool s_only = Cs && !C && !has_res && ((ldcs & 3) == 0) && ((reinterpret_cast<size_t>(Cs) & 7) == 0);
    const bool bias_ok = has_bias || true;   // bv[] is zero without a bias
    if (TM <= 3 && interior && bias_ok && c_only && has_res && vecR) {   // (TM = 4: the two residual buffers do not fit 256 registers)
#define EPI_KIND 0
#include "gemm_split_epilogue_fast.inc"
#undef EPI_KIND
    } else if (interior && s_only && !do_gelu) {
#define EPI_KIND 1
#include "gemm_split_epilogue_fast.inc"
#undef EPI_KIND
    } else if (interior && s_only && do_gelu) {
#define EPI_KIND 2
#include "gemm_split_epilogue_fast.inc"
#undef EPI_KIND
    } else if (interior && c_only && !has_res) {
#define EPI_KIND 3
#include "gemm_split_epilogue_fast.inc"
#undef EPI_KIND
    } else {
#include "gemm_split_epilogue.inc"
    }
  }
  CRA5_TRACE(3);
#ifdef CRA5_GEMM_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * 5 + 4] = clock64() - g_gemm_trace[blockIdx.x * 5 + 4];
#endif
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

template <int WM, int WN, int TM, int TN, bool LONGK, int STAGES = 2, int NPROD = 3, bool UE = false>
int launch(const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C, int ldc,
           unsigned short *Cs, long ldcs, const float *bias, const float *res, int ldr, int M, int N, int Kp,
           float wscale_inv, int flags, hipStream_t st, UnembedArgs ue = UnembedArgs{}) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_nt_split_kernel<WM, WN, TM, TN, LONGK, STAGES, NPROD, UE>), dim3(tiles_m * tiles_n),
                     dim3(WM * WN * 64), 0, st, A, lda, W, ldw, C, ldc, Cs, ldcs, bias, res, ldr, M, N, Kp, wscale_inv,
                     flags, tiles_n, ue);
  return (int)hipGetLastError();
}

// Overlap rows of the fused un-embed: x[c][10 i] = side[c][i - 1][1] (the ky = 10 row of the patch row above, first -
// the order of the overlap-add kernel this replaces) + side[c][i][0] (this patch row's ky = 0 row), de-normalised; row 0
// and row H - 1 have one contribution.  One thread per four pixels of an overlap row.
__global__ __launch_bounds__(256) void unembed_fixup_kernel(const float *__restrict__ side, const float *__restrict__ mean,
                                                            const float *__restrict__ stdv, float *__restrict__ x, int C,
                                                            int H, int W, int Hp) {
  const int w4 = W / 4;
  const size_t total = (size_t)C * (Hp + 1) * w4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int p4 = (int)(e % w4);
    const size_t t = e / w4;
    const int i = (int)(t % (Hp + 1)), c = (int)(t / (Hp + 1));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    bool have = false;
    if (i > 0) {
      v = *reinterpret_cast<const float4 *>(side + (((size_t)c * Hp + (i - 1)) * 2 + 1) * W + 4 * p4);
      have = true;
    }
    if (i < Hp) {
      const float4 b = *reinterpret_cast<const float4 *>(side + (((size_t)c * Hp + i) * 2 + 0) * W + 4 * p4);
      v = have ? make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w) : b;
    }
    if (mean) {
      const float m = mean[c], sd = stdv[c];
      v = make_float4(v.x * sd + m, v.y * sd + m, v.z * sd + m, v.w * sd + m);
    }
    *reinterpret_cast<float4 *>(x + ((size_t)c * H + (size_t)10 * i) * W + 4 * p4) = v;
  }
}

}  // namespace

static int gemm_dispatch(const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C, int ldc,
                         unsigned short *C_split, long ldcs, const float *bias, const float *res, int ldr, int M,
                         int N, int Kp, float wscale_inv, int flags, hipStream_t st) {
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
#define CRA5_GO(WM, WN, TM, TN, LK) \
  return launch<WM, WN, TM, TN, LK>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st)
// variant builds only: the product library reads no environment variable
#if defined(CRA5_TUNING_ENV) || defined(CRA5_GEMM_TRACE)
  static const int forced = [] {
    const char *e = getenv("CRA5_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
#else
  constexpr int forced = 0;
#endif
  const bool longk = Kp > 8192;
  // plain-f16 operands / output exist in the wide reduced-precision form only (the caller falls back to split rows)
  constexpr int PLAIN_ANY = CRA5_GEMM_A_PLAIN | CRA5_GEMM_W_PLAIN | CRA5_GEMM_OUT_PLAIN;
  if ((flags & PLAIN_ANY) && (!(flags & CRA5_GEMM_HI_ONLY) || longk || tiles128 < 256 || (Kp % 64))) return CRA5_ERR_ARG;
  if (longk) CRA5_GO(2, 2, 2, 2, true);
  if (flags & CRA5_GEMM_HI_ONLY) {   // reduced precision: one f16 MFMA per product
    const bool wide = (M >= 1024 && N >= 2048);
    if (tiles128 < 256)
      return launch<2, 2, 1, 1, false, 2, 1>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
    if (Kp % 64 == 0) {   // wide form: 64 k-values (the hi halves of two chunks) per k-step, the fp32-accurate mode's loop
      if (wide)
        return launch<2, 4, 4, 2, false, 2, 2>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
      return launch<2, 4, 3, 2, false, 2, 2>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
    }
    if (wide)
      return launch<2, 4, 4, 2, false, 2, 1>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
    return launch<2, 4, 3, 2, false, 2, 1>(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
  }
  int tile = forced;
  if (!tile) {
    // measured on MI355X (tools/gemm_bench.py, M = 10368): N = 1024 (proj, fc2, patch-embed):
    // 192x256 tiles = 216 blocks, one round on 256 CUs; N >= 3072: 256x256 tiles.
    if (tiles128 < 256) tile = 64;
    else if (M >= 1024 && N >= 2048) tile = 256;
    else if (M >= 1024 && N >= 512) tile = 192;
    else tile = 128;
  }
  // (a 4-wave 256 x 256 instantiation - one wave per SIMD, 128 x 128 per wave, 256 accumulator AGPRs, a third less
  // LDS fragment traffic per MFMA - compiles spill-free but measured 4-5 % SLOWER with hipcc's schedule: qkv 197 vs
  // 190 us, un-embed 1922 vs 1832; a single wave per SIMD needs a hand-placed MFMA / ds_read / LDS-DMA interleave)
  if (tile == 64) CRA5_GO(2, 2, 1, 1, false);
  if (tile == 192) CRA5_GO(2, 4, 3, 2, false);   // 192 x 256, 8 waves (2 x 4), 3 x 2 sub-tiles per wave
  if (tile == 256) CRA5_GO(2, 4, 4, 2, false);   // 256 x 256, 8 waves (2 x 4), 4 x 2 sub-tiles per wave
  CRA5_GO(2, 2, 2, 2, false);
#undef CRA5_GO
}

extern "C" int cra5_gemm_nt_split(const uint16_t *A, int lda_kp, const uint16_t *W, int ldw_kp, float *C, int ldc,
                                  uint16_t *C_split, int ldc_split_kp, const float *bias, const float *res, int ldr,
                                  int M, int N, int Kp, float wscale_inv, int flags, void *stream) {
  if (!A || !W || (!C && !C_split) || M <= 0 || N <= 0 || Kp <= 0 || (Kp % BK)) return CRA5_ERR_ARG;
  if (lda_kp < Kp || ldw_kp < Kp || (lda_kp % 32) || (ldw_kp % 32)) return CRA5_ERR_ARG;
  if (lda_kp >= (1 << 21) || ldw_kp >= (1 << 21)) return CRA5_ERR_ARG;   // 32-bit tile-relative DMA offsets (2 halves per k)
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_BIAS) && !bias) return CRA5_ERR_ARG;
  if ((flags & CRA5_EPI_RES) && !res) return CRA5_ERR_ARG;
  if (C_split && (ldc_split_kp % 32 || ldc_split_kp < N)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // row pitches in halves: a split row holds 2 halves per k, a plain row one (the caller states its pitch in halves)
  const bool a_plain = flags & CRA5_GEMM_A_PLAIN, w_plain = flags & CRA5_GEMM_W_PLAIN, o_plain = flags & CRA5_GEMM_OUT_PLAIN;
  const long lda = a_plain ? (long)lda_kp : 2L * lda_kp, ldw = w_plain ? (long)ldw_kp : 2L * ldw_kp;
  const long ldcs = o_plain ? (long)ldc_split_kp : 2L * ldc_split_kp;
  // Long reductions (patch-embed conv, K = 29 480): K is cut into chunks of <= 8192 whose
  // partial products are chained through the fp32 C matrix (C += A_i . W_i): a two-level
  // sum (each chunk accumulates in the MFMA accumulators, chunks add in fp32) without a
  // second accumulator register set, so the big tiles stay spill-free.
  if (Kp > 8192 && C && !C_split && !(flags & CRA5_EPI_GELU)) {
    const int nchunk = (Kp + 8191) / 8192;
    const int gran = ((flags & CRA5_GEMM_HI_ONLY) && Kp % 64 == 0) ? 64 : 32;   // the wide reduced-precision form steps by 64
    const int per = ((Kp / gran + nchunk - 1) / nchunk) * gran;
    for (int k0 = 0, i = 0; k0 < Kp; k0 += per, ++i) {
      const int kc = (Kp - k0 < per) ? Kp - k0 : per;
      const int f = ((i == 0) ? flags : CRA5_EPI_RES) | (flags & (CRA5_GEMM_HI_ONLY | CRA5_GEMM_A_PLAIN | CRA5_GEMM_W_PLAIN));
      const int rc = gemm_dispatch(A + (a_plain ? 1L : 2L) * k0, lda, W + (w_plain ? 1L : 2L) * k0, ldw, C, ldc, nullptr, 0, (i == 0) ? bias : nullptr,
                                   (i == 0) ? res : C, (i == 0) ? ldr : ldc, M, N, kc, wscale_inv, f, st);
      if (rc) return rc;
    }
    return 0;
  }
  return gemm_dispatch(A, lda, W, ldw, C, ldc, C_split, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
}

// Fixed-tile entry for the hyper-prior path (csrc/hyper.hip): 64 x 64 or 128 x 128 tiles chosen by the CALLER,
// never by CRA5_GEMM_TILE - the encode and the decode side must run the same kernel (bit-identical h_s).
extern "C" __attribute__((visibility("hidden"))) int cra5_internal_gemm_split_tile(
    const unsigned short *A, long lda, const unsigned short *W, long ldw, float *C, int ldc, unsigned short *Cs,
    long ldcs, const float *bias, const float *res, int ldr, int M, int N, int Kp, float wscale_inv, int flags,
    int tile, hipStream_t st) {
  if (tile == 64) return launch<2, 2, 1, 1, false>(A, lda, W, ldw, C, ldc, Cs, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
  return launch<2, 2, 2, 2, false>(A, lda, W, ldw, C, ldc, Cs, ldcs, bias, res, ldr, M, N, Kp, wscale_inv, flags, st);
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

extern "C" size_t cra5_unembed_side_bytes(int C, int H, int W, int kh, int kw, int sh, int sw) {
  if (C <= 0 || kh != 11 || kw != 10 || sh != 10 || sw != 10 || H < kh || W < kw || (H - kh) % sh || (W - kw) % sw) return 0;
  return (size_t)C * ((H - kh) / sh + 1) * 2 * W * sizeof(float);
}

extern "C" int cra5_gemm_nt_split_unembed(const uint16_t *A, int lda_kp, const uint16_t *Wt, int ldw_kp, float *x,
                                          float *side, size_t side_bytes, const float *mean, const float *stdv, int M,
                                          int Kp, float wscale_inv, int C, int H, int W, int kh, int kw, int sh, int sw,
                                          int hi_only, void *stream) {
  const size_t need = cra5_unembed_side_bytes(C, H, W, kh, kw, sh, sw);
  if (!need) return CRA5_ERR_ARG;                       // other geometries: plain GEMM + cra5_col2im_f32
  if (!A || !Wt || !x || !side || side_bytes < need || M <= 0 || Kp <= 0 || (Kp % BK) || Kp > 8192) return CRA5_ERR_ARG;
  if ((mean == nullptr) != (stdv == nullptr)) return CRA5_ERR_ARG;
  const int Hp = (H - kh) / sh + 1, Wp = (W - kw) / sw + 1, N = C * kh * kw;
  if (M != Hp * Wp || (W % 4)) return CRA5_ERR_ARG;
  if (lda_kp < Kp || ldw_kp < Kp || (lda_kp % 32) || (ldw_kp % 32) || lda_kp >= (1 << 21) || ldw_kp >= (1 << 21)) return CRA5_ERR_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)Wt & 15) || ((uintptr_t)x & 15) || ((uintptr_t)side & 15)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  UnembedArgs ue;
  ue.side = side;
  ue.mean = mean;
  ue.stdv = stdv;
  ue.H = H;
  ue.W = W;
  ue.Hp = Hp;
  ue.Wp = Wp;
  // hi_only: 0 = fp32-accurate; 1 = reduced precision on split operands; | 2: A is a plain f16 matrix, | 4: Wt is
  // (pitches then in halves, as for cra5_gemm_nt_split's CRA5_GEMM_A_PLAIN / _W_PLAIN)
  const bool a_plain = hi_only & 2, w_plain = hi_only & 4;
  if ((a_plain || w_plain) && (!(hi_only & 1) || (Kp % 64))) return CRA5_ERR_ARG;
  const long lda = a_plain ? (long)lda_kp : 2L * lda_kp, ldw = w_plain ? (long)ldw_kp : 2L * ldw_kp;
  int rc;
  if (hi_only && Kp % 64 == 0)
    rc = launch<2, 4, 4, 2, false, 2, 2, true>(A, lda, Wt, ldw, x, 0, nullptr, 0, nullptr, nullptr, 0, M, N, Kp, wscale_inv,
                                               CRA5_GEMM_HI_ONLY | (a_plain ? CRA5_GEMM_A_PLAIN : 0) | (w_plain ? CRA5_GEMM_W_PLAIN : 0), st, ue);
  else if (hi_only)
    rc = launch<2, 4, 4, 2, false, 2, 1, true>(A, lda, Wt, ldw, x, 0, nullptr, 0, nullptr, nullptr, 0, M, N, Kp, wscale_inv,
                                               CRA5_GEMM_HI_ONLY, st, ue);
  else
    rc = launch<2, 4, 4, 2, false, 2, 3, true>(A, lda, Wt, ldw, x, 0, nullptr, 0, nullptr, nullptr, 0, M, N, Kp, wscale_inv,
                                               0, st, ue);
  if (rc) return rc;
  const size_t total = (size_t)C * (Hp + 1) * (W / 4);
  size_t g = (total + 255) / 256;
  if (g > 65535) g = 65535;
  hipLaunchKernelGGL(unembed_fixup_kernel, dim3((unsigned)g), dim3(256), 0, st, side, mean, stdv, x, C, H, W, Hp);
  return (int)hipGetLastError();
}

#ifdef CRA5_GEMM_TRACE
extern "C" int cra5_debug_gemm_trace(unsigned long long *host, int n_blocks) {
  if (n_blocks > 8192) n_blocks = 8192;
  hipDeviceSynchronize();
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 5 * n_blocks);
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gemm_trace)) == hipSuccess) hipMemset(p, 0, sizeof(unsigned long long) * 5 * 8192);
  hipDeviceSynchronize();
  return rc;
}
#endif
