// HBM-bound kernels of the VAEformer path: LayerNorm, patch gather / overlap-add scatter
// (fused with the API's (de)normalisation), layout plumbing, and the two entropy-model
// kernels (GaussianConditional, EntropyBottleneck) + GDN.  All of them are streaming:
// coalesced 16-byte accesses, wave64 shuffles for reductions, no atomics (the h_s -> index
// path must be bit-reproducible between the encode and the decode side).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/cra5_amd.h"
#include "split.h"

CRA5_RANGE_TU(elementwise)
#ifdef CRA5_RANGE_CHECK
extern "C" {
int cra5_range_counts_gemm(unsigned long long *, int);
int cra5_range_counts_attn(unsigned long long *, int);
int cra5_range_counts_attn_f32(unsigned long long *, int);
int cra5_range_counts_hyper(unsigned long long *, int);
}
#endif

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, the row lives in registers (two-pass mean / variance,
// fixed butterfly reduction order -> deterministic).  vit_nlc.py:266,278 (eps = 1e-6).
// ---------------------------------------------------------------------------------------
template <int V4>  // float4 per lane (D <= V4 * 256)
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int ldx,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float *__restrict__ y,
                                                        int ldy, unsigned short *__restrict__ ys, int Kp, int rows,
                                                        int D, float eps, int ys_plain) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + (size_t)row * ldx;
  float4 v[V4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D) v[i] = *reinterpret_cast<const float4 *>(xr + c);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  float *yr = y ? y + (size_t)row * ldy : nullptr;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
      const float4 b = *reinterpret_cast<const float4 *>(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (y) *reinterpret_cast<float4 *>(yr + c) = o;
      if (ys) {   // (row pitch 2 * Kp halves either way: a plain row is the first half of a split row)
        if (ys_plain) cra5_store_plain4(ys + (size_t)row * 2 * Kp, c, o.x, o.y, o.z, o.w);
        else cra5_store_split4(ys + (size_t)row * 2 * Kp, c, o.x, o.y, o.z, o.w);
      }
    }
  }
  if (ys)  // zero the K padding of the split row (D..Kp)
    for (int c = D + lane; c < Kp; c += 64) {
      if (ys_plain) cra5_store_plain(ys + (size_t)row * 2 * Kp, c, 0.f);
      else cra5_store_split(ys + (size_t)row * 2 * Kp, c, 0.f);
    }
}

// ---------------------------------------------------------------------------------------
// im2col (patch gather) with fused normalisation.  One thread per output float; consecutive
// threads walk the K (= c, i, j) axis of one token -> coalesced stores.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                     const float *__restrict__ stdv, float *__restrict__ cols,
                                                     unsigned short *__restrict__ cols_s, int C, int H, int W,
                                                     int kh, int kw, int sh, int sw, int Hp, int Wp, int ldk, int plain) {
  const int K = C * kh * kw;
  const size_t total = (size_t)Hp * Wp * K;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int tok = (int)(e / K);
    const int k = (int)(e - (size_t)tok * K);
    const int ph = tok / Wp, pw = tok - ph * Wp;
    const int c = k / (kh * kw);
    const int ij = k - c * kh * kw;
    const int i = ij / kw, j = ij - i * kw;
    float v = x[((size_t)c * H + (ph * sh + i)) * W + (pw * sw + j)];
    if (mean) v = (v - mean[c]) / stdv[c];
    if (cols) cols[(size_t)tok * ldk + k] = v;
    if (cols_s) {
      if (plain) cra5_store_plain(cols_s + (size_t)tok * 2 * ldk, k, v);
      else cra5_store_split(cols_s + (size_t)tok * 2 * ldk, k, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// col2im (ConvTranspose2d overlap-add) with fused de-normalisation.  Gather formulation:
// one thread per output pixel sums its <= ceil(kh/sh)*ceil(kw/sw) contributions in a fixed
// order -> no atomics, deterministic.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col2im_kernel(const float *__restrict__ cols, const float *__restrict__ mean,
                                                     const float *__restrict__ stdv, float *__restrict__ x, int C,
                                                     int H, int W, int kh, int kw, int sh, int sw, int Hp, int Wp,
                                                     int ldn) {
  const size_t total = (size_t)C * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(e % W);
    const size_t t = e / W;
    const int row = (int)(t % H);
    const int c = (int)(t / H);
    // patches ph with 0 <= row - ph*sh < kh
    int ph_hi = row / sh;
    if (ph_hi > Hp - 1) ph_hi = Hp - 1;
    int ph_lo = (row - kh + sh) / sh;  // ceil((row-kh+1)/sh) for row-kh+1 >= 0
    if (row - kh + 1 < 0) ph_lo = 0;
    int pw_hi = col / sw;
    if (pw_hi > Wp - 1) pw_hi = Wp - 1;
    int pw_lo = (col - kw + sw) / sw;
    if (col - kw + 1 < 0) pw_lo = 0;
    float acc = 0.f;
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
      const int i = row - ph * sh;
      for (int pw = pw_lo; pw <= pw_hi; ++pw) {
        const int j = col - pw * sw;
        acc += cols[(size_t)(ph * Wp + pw) * ldn + (c * kh + i) * kw + j];
      }
    }
    if (mean) acc = acc * stdv[c] + mean[c];
    x[e] = acc;
  }
}


// ---------------------------------------------------------------------------------------
// LDS-tiled patch gather / overlap-add for the ERA5 geometry (kw == sw: patches do not overlap
// along W; kh = sh + 1: one shared row between vertically adjacent patches).
// One block = (patch row ph, 16 consecutive patches, 8 channels): the 8 x kh image-row segments
// of 16*kw floats (640-byte coalesced runs) are staged once in LDS; every output token then
// writes 8*kh*kw consecutive K entries (3.5 KB runs).  HBM-bound: 1.11 GB in, 1.22 GB out.
// ---------------------------------------------------------------------------------------
constexpr int TILE_TOK = 16;
// channels per block: 8 -> 2 took the gather from 44 % to 54 % and the scatter from 31 % to 63 % of the
// HBM peak (LDS per block 56 KB -> 14 KB: 2 -> 8 resident blocks per CU; tools/mem_bench.py).  Must stay
// even: the K offset c0*KH*KW of a block has to be a multiple of 4 floats for the 16-byte accesses.
constexpr int TILE_CH = 2;      // gather (im2col)
constexpr int TILE_CH_S = 2;    // scatter (col2im)

// LDS image of both kernels: token-major [16 tokens][8*KH*KW (+4 pad)] = the GEMM-side layout,
// so the GEMM-side accesses are contiguous ds_read/write_b128 + 16-byte global accesses, and the
// (c, i, j) <-> (row, column) transposition is done with scalar LDS accesses on the image side.
template <int KH, int KW>
__global__ __launch_bounds__(256) void im2col_tiled_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                           const float *__restrict__ stdv, float *__restrict__ cols,
                                                           unsigned short *__restrict__ cols_s, int C, int H, int W,
                                                           int sh, int Hp, int Wp, int ldk, int plain) {
  constexpr int SEG = TILE_TOK * KW;             // floats per staged image-row segment
  constexpr int TS = TILE_CH * KH * KW + 4;      // LDS token stride (floats)
  static_assert((TILE_CH * KH * KW) % 4 == 0, "a block's K offset must stay 16-byte aligned");
  __shared__ __attribute__((aligned(16))) float tile[TILE_TOK * TS];
  // block order: channel chunk fastest -> consecutive blocks sweep the K axis of the same 16
  // tokens (the 118 KB token rows of the column matrix are streamed front to back)
  const int tiles_w = Wp / TILE_TOK;
  const int n_cc = (C + TILE_CH - 1) / TILE_CH;
  const int cc = blockIdx.x % n_cc;
  const int pwt = (blockIdx.x / n_cc) % tiles_w;
  const int ph = blockIdx.x / (n_cc * tiles_w);
  const int c0 = cc * TILE_CH;
  const int nc = min(TILE_CH, C - c0);
  const int col0 = pwt * SEG;
  // ---- stage: image rows (c, i) in 16-byte pieces -> tile[t][(c*KH + i)*KW + j] ------------------
  const int n4 = nc * KH * (SEG / 4);
  constexpr int UNR = 7;   // independent 16-byte loads in flight per thread (memory-level parallelism)
  for (int e0 = threadIdx.x; e0 < n4; e0 += 256 * UNR) {
    float4 v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int e = min(e0 + q * 256, n4 - 1);
      const int row = e / (SEG / 4), p4 = e - row * (SEG / 4);
      const int c = row / KH, i = row - c * KH;
      v[q] = *reinterpret_cast<const float4 *>(x + ((size_t)(c0 + c) * H + (ph * sh + i)) * W + col0 + p4 * 4);
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int e = e0 + q * 256;
      if (e < n4) {
        const int row = e / (SEG / 4), p4 = e - row * (SEG / 4);   // row = c*KH + i
        const int c = row / KH;
        float vv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
        if (mean) {
          const float m = mean[c0 + c], sd = stdv[c0 + c];
#pragma unroll
          for (int u = 0; u < 4; ++u) vv[u] = (vv[u] - m) / sd;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int col = p4 * 4 + u;
          const int t = col / KW, j = col - t * KW;
          tile[t * TS + row * KW + j] = vv[u];
        }
      }
    }
  }
  __syncthreads();
  // ---- emit: contiguous K runs per token ----------------------------------------------------------
  const int kpt = nc * KH * KW;                  // K entries per token in this block
  const int q4 = (kpt + 3) / 4;
  const int kbase = c0 * KH * KW;                // multiple of 4 (TILE_CH * KH * KW = 220)
  for (int e = threadIdx.x; e < TILE_TOK * q4; e += 256) {
    const int t = e / q4, k4 = (e - t * q4) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(tile + t * TS + k4);
    const size_t tok = (size_t)ph * Wp + pwt * TILE_TOK + t;
    if (k4 + 4 <= kpt) {
      if (cols) *reinterpret_cast<float4 *>(cols + tok * ldk + kbase + k4) = v;
      if (cols_s) {   // (plain: the reduced-precision mode's rows - element k at half k of the 2 * ldk-halves row)
        if (plain) cra5_store_plain4(cols_s + tok * 2 * ldk, kbase + k4, v.x, v.y, v.z, v.w);
        else cra5_store_split4(cols_s + tok * 2 * ldk, kbase + k4, v.x, v.y, v.z, v.w);
      }
    } else {   // ragged end of the last channel chunk (e.g. 159 = 19*8 + 7 channels)
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int u = 0; k4 + u < kpt; ++u) {
        if (cols) cols[tok * ldk + kbase + k4 + u] = vv[u];
        if (cols_s) {
          if (plain) cra5_store_plain(cols_s + tok * 2 * ldk, kbase + k4 + u, vv[u]);
          else cra5_store_split(cols_s + tok * 2 * ldk, kbase + k4 + u, vv[u]);
        }
      }
    }
  }
  // plain rows: the K padding (C*KH*KW .. ldk) is written every time - the same workspace holds split rows in the
  // fp32-accurate mode, whose lo planes lie where a plain row has its padding
  if (plain && cols_s && cc == n_cc - 1) {
    const int K = C * KH * KW, npad = ldk - K;
    for (int e = threadIdx.x; e < TILE_TOK * npad; e += 256) {
      const int t = e / npad, k = K + (e - t * npad);
      cols_s[((size_t)ph * Wp + pwt * TILE_TOK + t) * 2 * ldk + k] = 0;
    }
  }
}

template <int KH, int KW>
__global__ __launch_bounds__(256) void col2im_tiled_kernel(const float *__restrict__ cols, const float *__restrict__ mean,
                                                           const float *__restrict__ stdv, float *__restrict__ x, int C,
                                                           int H, int W, int sh, int Hp, int Wp, int ldn) {
  constexpr int SEG = TILE_TOK * KW;
  constexpr int TS = TILE_CH_S * KH * KW + 4;
  static_assert((TILE_CH_S * KH * KW) % 4 == 0, "a block's K offset must stay 16-byte aligned");
  __shared__ __attribute__((aligned(16))) float tile[TILE_TOK * TS];
  // the i = sh row of the patch row above lands on this block's first output row (overlap-add);
  // KH - sh == 1 is checked by the launcher
  __shared__ __attribute__((aligned(16))) float above[TILE_CH_S * SEG];
  // block order: channel chunk fastest -> consecutive blocks sweep the K axis of the same 16
  // tokens (the 118 KB token rows of the column matrix are streamed front to back)
  const int tiles_w = Wp / TILE_TOK;
  const int n_cc = (C + TILE_CH_S - 1) / TILE_CH_S;
  const int cc = blockIdx.x % n_cc;
  const int pwt = (blockIdx.x / n_cc) % tiles_w;
  const int ph = blockIdx.x / (n_cc * tiles_w);
  const int c0 = cc * TILE_CH_S;
  const int nc = min(TILE_CH_S, C - c0);
  const int kpt = nc * KH * KW, q4 = (kpt + 3) / 4, kbase = c0 * KH * KW;
  // ---- stage this patch row: 3.5 KB token runs, straight copy ------------------------------------
  constexpr int UNR = 7;
  const int nq = TILE_TOK * q4;
  for (int e0 = threadIdx.x; e0 < nq; e0 += 256 * UNR) {
    float4 v[UNR];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int e = min(e0 + q * 256, nq - 1);
      const int t = e / q4, k4 = min((e - t * q4) * 4, ((kpt - 4) / 4) * 4);   // in-bounds 16-byte load
      const size_t tok = (size_t)ph * Wp + pwt * TILE_TOK + t;
      v[q] = *reinterpret_cast<const float4 *>(cols + tok * ldn + kbase + k4);
    }
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int e = e0 + q * 256;
      if (e < nq) {
        const int t = e / q4, k4 = (e - t * q4) * 4;
        if (k4 + 4 <= kpt) {
          *reinterpret_cast<float4 *>(tile + t * TS + k4) = v[q];
        } else {   // ragged end of the last channel chunk: re-read the tail scalars
          const size_t tok = (size_t)ph * Wp + pwt * TILE_TOK + t;
          for (int u = 0; k4 + u < kpt; ++u) tile[t * TS + k4 + u] = cols[tok * ldn + kbase + k4 + u];
        }
      }
    }
  }
  if (ph > 0) {   // TILE_CH_S * SEG / 256 = 5 independent scalar loads per thread, issued together
    constexpr int AU = (TILE_CH_S * SEG + 255) / 256;
    float av[AU];
#pragma unroll
    for (int q = 0; q < AU; ++q) {
      const int e = min((int)threadIdx.x + q * 256, nc * SEG - 1);
      const int c = e / SEG, p = e - c * SEG;
      const int t = p / KW, j = p - t * KW;
      const size_t tok = (size_t)(ph - 1) * Wp + pwt * TILE_TOK + t;
      av[q] = cols[tok * ldn + ((size_t)(c0 + c) * KH + sh) * KW + j];
    }
#pragma unroll
    for (int q = 0; q < AU; ++q) {
      const int e = threadIdx.x + q * 256;
      if (e < nc * SEG) above[e] = av[q];
    }
  }
  __syncthreads();
  // ---- emit image rows: i in [0, sh) (+ the last row i = sh for the last patch row) ----------------
  const int rows_out = (ph == Hp - 1) ? KH : sh;
  const int n4 = nc * rows_out * (SEG / 4);
  for (int e = threadIdx.x; e < n4; e += 256) {
    const int row = e / (SEG / 4), p4 = e - row * (SEG / 4);
    const int c = row / rows_out, i = row - c * rows_out;
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = p4 * 4 + u;
      const int t = col / KW, j = col - t * KW;
      vv[u] = tile[t * TS + (c * KH + i) * KW + j];
    }
    if (ph > 0 && i == 0) {   // same order as the gather kernel: upper patch row first, then this one
      const float4 a = *reinterpret_cast<const float4 *>(above + c * SEG + p4 * 4);
      vv[0] = a.x + vv[0];
      vv[1] = a.y + vv[1];
      vv[2] = a.z + vv[2];
      vv[3] = a.w + vv[3];
    }
    if (mean) {
      const float m = mean[c0 + c], sd = stdv[c0 + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) vv[u] = vv[u] * sd + m;
    }
    *reinterpret_cast<float4 *>(x + ((size_t)(c0 + c) * H + (ph * sh + i)) * W + pwt * SEG + p4 * 4) =
        make_float4(vv[0], vv[1], vv[2], vv[3]);
  }
}

// ---------------------------------------------------------------------------------------
// 32x32 LDS-tiled transpose
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int ld_in,
                                                        float *__restrict__ out, int ld_out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = in[(size_t)r * ld_in + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) out[(size_t)c * ld_out + r] = tile[tx][ty + 8 * k];
  }
}

// Finiteness probe of the range guard (cra5_amd/vaeformer.py: _range_guard): CRA5_PROBE_PARTIALS partial sums of
// x[0], x[stride], x[2 stride], ... - one per block, written (not accumulated: no memset, no atomics, deterministic).  A
// sum is non-finite as soon as one addend is (fp32 sums of O(1e7) bounded activations do not overflow): the host tests
// the partials it copies back with the phase's other results.  Replaces the torch reductions / stack / cat kernels of
// rounds 1-4 on the frame path.
__global__ __launch_bounds__(256) void probe_sums_kernel(const float *__restrict__ x, size_t n, size_t stride,
                                                         float *__restrict__ partials) {
  const size_t cnt = (n + stride - 1) / stride;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x)
    acc += x[i * stride];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// 'b h w (p1 p2 c) -> b c (h p1) (w p2)'
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float *__restrict__ lin, float *__restrict__ out,
                                                            int Hz, int Wz, int p1, int p2, int Cout) {
  const int Wo = Wz * p2, Ho = Hz * p1;
  const size_t total = (size_t)Cout * Ho * Wo;
  const int F = p1 * p2 * Cout;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int wo = (int)(e % Wo);
    const size_t t = e / Wo;
    const int ho = (int)(t % Ho);
    const int c = (int)(t / Ho);
    const int hz = ho / p1, a = ho - hz * p1;
    const int wz = wo / p2, b = wo - wz * p2;
    out[e] = lin[(size_t)(hz * Wz + wz) * F + (a * p2 + b) * Cout + c];
  }
}

// ---------------------------------------------------------------------------------------
// CNN zoo plumbing (models/utils.py:128-147 conv / deconv: kernel k, stride s, padding k/2,
// output_padding s-1): Conv2d = zero-padded patch gather + GEMM, ConvTranspose2d = GEMM + the gather
// form of the overlap-add (every output pixel sums its <= ceil(k/s)^2 contributions in a fixed order:
// deterministic, no atomics).  Small images, latency-irrelevant: one thread per element.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_im2col_kernel(const float *__restrict__ x, unsigned short *__restrict__ cols_s,
                                                          int C, int H, int W, int kh, int kw, int sh, int sw, int ph,
                                                          int pw, int Ho, int Wo, int Kp) {
  const int K = C * kh * kw;
  const size_t total = (size_t)Ho * Wo * Kp;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % Kp);
    const size_t tok = e / Kp;
    float v = 0.f;
    if (k < K) {
      const int j = k % kw, i = (k / kw) % kh, c = k / (kw * kh);
      const int yo = (int)(tok / Wo), xo = (int)(tok % Wo);
      const int y = yo * sh - ph + i, xx = xo * sw - pw + j;
      if (y >= 0 && y < H && xx >= 0 && xx < W) v = x[((size_t)c * H + y) * W + xx];
    }
    cra5_store_split(cols_s + tok * 2 * Kp, k, v);
  }
}

// out[co][y][x] = bias[co] + sum_{i, j} cols[(yi, xi)][(co kh + i) kw + j],  y = yi s - p + i, x = xi s - p + j
__global__ __launch_bounds__(256) void deconv_col2im_kernel(const float *__restrict__ cols, const float *__restrict__ bias,
                                                            float *__restrict__ out, int Cout, int Hi, int Wi, int kh,
                                                            int kw, int sh, int sw, int ph, int pw, int Ho, int Wo,
                                                            int ldn) {
  const size_t total = (size_t)Cout * Ho * Wo;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int xo = (int)(e % Wo);
    const size_t t = e / Wo;
    const int yo = (int)(t % Ho), co = (int)(t / Ho);
    float acc = bias ? bias[co] : 0.f;
    for (int i = (yo + ph) % sh; i < kh; i += sh) {
      const int yi = (yo + ph - i) / sh;
      if (yo + ph - i < 0 || yi >= Hi) continue;
      for (int j = (xo + pw) % sw; j < kw; j += sw) {
        const int xi = (xo + pw - j) / sw;
        if (xo + pw - j < 0 || xi >= Wi) continue;
        acc += cols[(size_t)(yi * Wi + xi) * ldn + (co * kh + i) * kw + j];
      }
    }
    out[e] = acc;
  }
}

__global__ __launch_bounds__(256) void unary_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n, int op,
                                                    float slope) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = x[e];
    y[e] = (op == 1) ? __builtin_fabsf(v) : ((v >= 0.f) ? v : v * slope);   // 0: relu / leaky relu, 1: abs
  }
}

// ---------------------------------------------------------------------------------------
// GaussianConditional (entropy_models.py:645-685).  Phi(u) = 0.5 * erfc(-u / sqrt 2).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float phi(float u) { return 0.5f * erfcf(-0.70710678118654752440f * u); }

__global__ __launch_bounds__(256) void gaussian_conditional_kernel(
    const float *__restrict__ y, const int32_t *__restrict__ sym_in, const float *__restrict__ scales,
    const float *__restrict__ means, const float *__restrict__ table, int n_table, float scale_bound,
    float lik_bound, int32_t *__restrict__ idx, int32_t *__restrict__ sym, float *__restrict__ y_hat,
    float *__restrict__ lik, size_t n) {
  __shared__ float tb[256];
  for (int i = threadIdx.x; i < n_table; i += blockDim.x) tb[i] = table[i];
  __syncthreads();
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float mu = means[e];
    const float s = fmaxf(scales[e], scale_bound);  // LowerBound
    float q;                                        // quantised residual
    if (y) q = rintf(y[e] - mu);                    // torch.round = half-to-even
    else q = (float)sym_in[e];
    if (idx) {
      int id = n_table - 1;
      for (int t = 0; t < n_table - 1; ++t) id -= (s <= tb[t]) ? 1 : 0;
      idx[e] = id;
    }
    if (sym) sym[e] = (int32_t)q;
    const float yh = q + mu;
    if (y_hat) y_hat[e] = yh;
    if (lik) {
      const float v = fabsf(yh - mu);
      const float up = phi((0.5f - v) / s);
      const float lo = phi((-0.5f - v) / s);
      lik[e] = fmaxf(up - lo, lik_bound);
    }
  }
}

// decode side on compact records: uint8 CDF indexes out (the D2H copy the host decoder waits for: 2.65 instead of
// 10.6 MB per frame), int16 symbols in (5.3 instead of 10.6 MB); same arithmetic as the kernel above
__global__ __launch_bounds__(256) void gaussian_conditional_compact_kernel(
    const float *__restrict__ scales, const float *__restrict__ means, const float *__restrict__ table, int n_table,
    float scale_bound, const int16_t *__restrict__ sym16_in, uint8_t *__restrict__ idx8, float *__restrict__ y_hat,
    size_t n) {
  __shared__ float tb[256];
  for (int i = threadIdx.x; i < n_table; i += blockDim.x) tb[i] = table[i];
  __syncthreads();
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    if (idx8) {
      const float s = fmaxf(scales[e], scale_bound);  // LowerBound
      int id = n_table - 1;
      for (int t = 0; t < n_table - 1; ++t) id -= (s <= tb[t]) ? 1 : 0;
      idx8[e] = (uint8_t)id;
    }
    if (y_hat) y_hat[e] = (float)sym16_in[e] + means[e];
  }
}

// ---------------------------------------------------------------------------------------
// Symbol -> (start, range, escape payload) against the quantised CDF tables, on the device
// (rans_interface.cpp:121-150; the host encoder then only updates its state).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolve_symbols_kernel(
    const int32_t *__restrict__ sym, const int32_t *__restrict__ idx, size_t n, const int32_t *__restrict__ cdfs,
    int n_cdfs, int stride, const int32_t *__restrict__ sizes, const int32_t *__restrict__ offsets,
    uint32_t *__restrict__ sr, uint32_t *__restrict__ raw, uint8_t *__restrict__ esc) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = idx[e];
    if (ci < 0 || ci >= n_cdfs || sizes[ci] < 2 || sizes[ci] > stride) {
      sr[e] = 0u;
      raw[e] = 0u;
      esc[e] = 255;
      continue;
    }
    const int32_t *cdf = cdfs + (size_t)ci * stride;
    const int max_value = sizes[ci] - 2;
    int value = sym[e] - offsets[ci];
    uint32_t r = 0u;
    if (value < 0) {
      r = (uint32_t)(-2 * value - 1);
      value = max_value;
    } else if (value >= max_value) {
      r = (uint32_t)(2 * (value - max_value));
      value = max_value;
    }
    const uint32_t start = (uint32_t)cdf[value] & 0xFFFFu;
    const uint32_t range = (uint32_t)(cdf[value + 1] - cdf[value]) & 0xFFFFu;
    sr[e] = start | (range << 16);
    raw[e] = r;
    uint8_t ec = 0;
    if (value == max_value) {
      int nn = 0;
      while (nn < 8 && (r >> (4 * nn)) != 0u) ++nn;
      ec = (uint8_t)(nn + 1);
    }
    esc[e] = ec;
  }
}

// The same resolve step writing COMPACT records: rec16 = 0 for a regular symbol, (1 + payload nibbles) << 12 | payload
// for an escape whose payload fits 12 bits (|symbol| up to ~2000 beyond its table row), 0xFFFF + *overflow = 1 otherwise
// (the caller then takes the 32-bit records of resolve_symbols_kernel): 6 instead of 9 bytes per latent over PCIe.
__global__ __launch_bounds__(256) void resolve_symbols_compact_kernel(
    const int32_t *__restrict__ sym, const int32_t *__restrict__ idx, size_t n, const int32_t *__restrict__ cdfs,
    int n_cdfs, int stride, const int32_t *__restrict__ sizes, const int32_t *__restrict__ offsets,
    uint32_t *__restrict__ sr, uint16_t *__restrict__ rec16, int32_t *__restrict__ overflow) {
  bool bad = false;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int ci = idx[e];
    if (ci < 0 || ci >= n_cdfs || sizes[ci] < 2 || sizes[ci] > stride) {
      sr[e] = 0u;
      rec16[e] = 0xFFFFu;
      bad = true;
      continue;
    }
    const int32_t *cdf = cdfs + (size_t)ci * stride;
    const int max_value = sizes[ci] - 2;
    int value = sym[e] - offsets[ci];
    uint32_t r = 0u;
    if (value < 0) {
      r = (uint32_t)(-2 * value - 1);
      value = max_value;
    } else if (value >= max_value) {
      r = (uint32_t)(2 * (value - max_value));
      value = max_value;
    }
    const uint32_t start = (uint32_t)cdf[value] & 0xFFFFu;
    const uint32_t range = (uint32_t)(cdf[value + 1] - cdf[value]) & 0xFFFFu;
    sr[e] = start | (range << 16);
    uint16_t rec = 0;
    if (value == max_value) {
      if (r >= 4096u) {
        rec = 0xFFFFu;
        bad = true;
      } else {
        int nn = 0;
        while (nn < 3 && (r >> (4 * nn)) != 0u) ++nn;
        rec = (uint16_t)(((uint32_t)(nn + 1) << 12) | r);
      }
    }
    rec16[e] = rec;
  }
  if (bad) atomicOr(overflow, 1);
}

// ---------------------------------------------------------------------------------------
// EntropyBottleneck (entropy_models.py:434-510).  Per-channel parameter block (58 floats):
//   [ 0: 3) sp(M0)   [ 3: 6) b0   [ 6: 9) th(f0)
//   [ 9:18) sp(M1)   [18:21) b1   [21:24) th(f1)
//   [24:33) sp(M2)   [33:36) b2   [36:39) th(f2)
//   [39:48) sp(M3)   [48:51) b3   [51:54) th(f3)
//   [54:57) sp(M4)   [57]    b4
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float eb_logits(const float *p, float v) {
  float a[3], b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[i] = p[i] * v + p[3 + i];
    a[i] += p[6 + i] * tanhf(a[i]);
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float *m = p + 9 + 15 * l;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float t = m[3 * i + 0] * a[0];
      t += m[3 * i + 1] * a[1];
      t += m[3 * i + 2] * a[2];
      t += m[9 + i];
      b[i] = t + m[12 + i] * tanhf(t);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = b[i];
  }
  float t = p[54] * a[0];
  t += p[55] * a[1];
  t += p[56] * a[2];
  return t + p[57];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void entropy_bottleneck_kernel(
    const float *__restrict__ z, const int32_t *__restrict__ sym_in, const float *__restrict__ medians,
    const float *__restrict__ params, float lik_bound, int32_t *__restrict__ sym, float *__restrict__ z_hat,
    float *__restrict__ lik, int C, int n_per_ch) {
  const size_t total = (size_t)C * n_per_ch;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e / n_per_ch);
    const float med = medians[c];
    float q;
    if (z) q = rintf(z[e] - med);
    else q = (float)sym_in[e];
    if (sym) sym[e] = (int32_t)q;
    const float zh = q + med;
    if (z_hat) z_hat[e] = zh;
    if (lik) {
      const float *p = params + 58 * c;
      const float lo = eb_logits(p, zh - 0.5f);
      const float up = eb_logits(p, zh + 0.5f);
      lik[e] = fmaxf(sigmoidf_(up) - sigmoidf_(lo), lik_bound);
    }
  }
}

// ---------------------------------------------------------------------------------------
// GDN / IGDN (layers/gdn.py:76-92): per pixel a C x C mat-vec on x^2.  One thread per
// (b, pixel, out-channel); the x^2 vector of the pixel is shared through LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gdn_kernel(const float *__restrict__ x, const float *__restrict__ beta,
                                                  const float *__restrict__ gamma, float *__restrict__ y, int B,
                                                  int C, int HW, int inverse) {
  extern __shared__ float sq[];  // [PIX][C]
  const int PIX = 256 / 64 * 16; // 64 pixels per block
  const int p0 = blockIdx.x * PIX;
  const int b = blockIdx.y;
  const float *xb = x + (size_t)b * C * HW;
  float *yb = y + (size_t)b * C * HW;
  for (int e = threadIdx.x; e < PIX * C; e += blockDim.x) {
    const int c = e / PIX, p = e - c * PIX;
    const float v = (p0 + p < HW) ? xb[(size_t)c * HW + p0 + p] : 0.f;
    sq[p * C + c] = v * v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PIX * C; e += blockDim.x) {
    const int c = e / PIX, p = e - c * PIX;
    if (p0 + p >= HW) continue;
    float n = beta[c];
    const float *gr = gamma + (size_t)c * C;
    const float *s = sq + p * C;
    for (int j = 0; j < C; ++j) n += gr[j] * s[j];
    const float xv = xb[(size_t)c * HW + p0 + p];
    yb[(size_t)c * HW + p0 + p] = inverse ? xv * sqrtf(n) : xv * (1.0f / sqrtf(n));
  }
}

inline int grid_for(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  const size_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride the rest
  return (int)(g < cap ? (g ? g : 1) : cap);
}

}  // namespace

extern "C" {

int cra5_layernorm_f32(const float *x, int ldx, const float *gamma, const float *beta, float *y, int ldy,
                       uint16_t *y_split, int split_kp, int rows, int D, float eps, int split_plain, void *stream) {
  if (!x || !gamma || !beta || (!y && !y_split) || rows <= 0 || D <= 0 || (D & 3) || (ldx & 3) || D > 2048)
    return CRA5_ERR_ARG;
  if (y && (ldy & 3)) return CRA5_ERR_ARG;
  if (y_split && (split_kp < D || split_kp % 32)) return CRA5_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
#define CRA5_LN(V) hipLaunchKernelGGL(layernorm_kernel<V>, grid, block, 0, st, x, ldx, gamma, beta, y, ldy, y_split, split_kp, rows, D, eps, split_plain)
  if (D <= 256) CRA5_LN(1);
  else if (D <= 512) CRA5_LN(2);
  else if (D <= 1024) CRA5_LN(4);
  else CRA5_LN(8);
#undef CRA5_LN
  return (int)hipGetLastError();
}

int cra5_im2col_f32(const float *x, const float *mean, const float *stdv, float *cols, uint16_t *cols_split, int C,
                    int H, int W, int kh, int kw, int sh, int sw, int Hp, int Wp, int ldk, int split_plain, void *stream) {
  if (!x || (!cols && !cols_split) || C <= 0 || ldk < C * kh * kw) return CRA5_ERR_ARG;
  if (cols_split && (ldk % 32)) return CRA5_ERR_ARG;
  if ((Hp - 1) * sh + kh > H || (Wp - 1) * sw + kw > W) return CRA5_ERR_ARG;
  if ((mean == nullptr) != (stdv == nullptr)) return CRA5_ERR_ARG;
  // ERA5 patch geometry: LDS-tiled streaming kernel
  if (kh == 11 && kw == 10 && sw == 10 && sh == 10 && Wp % TILE_TOK == 0 && (W % 4) == 0 && (ldk % 4) == 0 &&
      ((uintptr_t)x & 15) == 0) {
    const int blocks = (Wp / TILE_TOK) * Hp * ((C + TILE_CH - 1) / TILE_CH);
    hipLaunchKernelGGL((im2col_tiled_kernel<11, 10>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, mean, stdv,
                       cols, cols_split, C, H, W, sh, Hp, Wp, ldk, split_plain);
    return (int)hipGetLastError();
  }
  if (split_plain && ldk != C * kh * kw) return CRA5_ERR_ARG;   // (the generic kernel does not re-zero a plain row's padding)
  const size_t total = (size_t)Hp * Wp * C * kh * kw;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, cols,
                     cols_split, C, H, W, kh, kw, sh, sw, Hp, Wp, ldk, split_plain);
  return (int)hipGetLastError();
}

int cra5_col2im_f32(const float *cols, const float *mean, const float *stdv, float *x, int C, int H, int W, int kh,
                    int kw, int sh, int sw, int Hp, int Wp, int ldn, void *stream) {
  if (!x || !cols || C <= 0 || ldn < C * kh * kw) return CRA5_ERR_ARG;
  if ((Hp - 1) * sh + kh != H || (Wp - 1) * sw + kw != W) return CRA5_ERR_ARG;
  if ((mean == nullptr) != (stdv == nullptr)) return CRA5_ERR_ARG;
  if (kh == 11 && kw == 10 && sw == 10 && sh == 10 && Wp % TILE_TOK == 0 && (W % 4) == 0 && (ldn % 4) == 0 &&
      ((uintptr_t)cols & 15) == 0 && ((uintptr_t)x & 15) == 0) {
    const int blocks = (Wp / TILE_TOK) * Hp * ((C + TILE_CH_S - 1) / TILE_CH_S);
    hipLaunchKernelGGL((col2im_tiled_kernel<11, 10>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, cols, mean,
                       stdv, x, C, H, W, sh, Hp, Wp, ldn);
    return (int)hipGetLastError();
  }
  const size_t total = (size_t)C * H * W;
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, cols, mean, stdv, x, C,
                     H, W, kh, kw, sh, sw, Hp, Wp, ldn);
  return (int)hipGetLastError();
}

int cra5_probe_sums_f32(const float *x, size_t n, size_t stride, float *partials, void *stream) {
  if (!x || !partials || n == 0 || stride == 0) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(probe_sums_kernel, dim3(CRA5_PROBE_PARTIALS), dim3(256), 0, (hipStream_t)stream, x, n, stride, partials);
  return (int)hipGetLastError();
}

int cra5_transpose_f32(const float *in, int ld_in, float *out, int ld_out, int rows, int cols, void *stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return CRA5_ERR_ARG;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
  hipLaunchKernelGGL(transpose_kernel, grid, block, 0, (hipStream_t)stream, in, ld_in, out, ld_out, rows, cols);
  return (int)hipGetLastError();
}

int cra5_pixel_shuffle_f32(const float *lin, float *out, int Hz, int Wz, int p1, int p2, int Cout, void *stream) {
  if (!lin || !out || Hz <= 0 || Wz <= 0 || p1 <= 0 || p2 <= 0 || Cout <= 0) return CRA5_ERR_ARG;
  const size_t total = (size_t)Cout * Hz * p1 * Wz * p2;
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, lin, out, Hz, Wz,
                     p1, p2, Cout);
  return (int)hipGetLastError();
}

int cra5_conv_im2col_f32(const float *x, uint16_t *cols_split, int C, int H, int W, int kh, int kw, int sh, int sw, int ph,
                         int pw, int Ho, int Wo, int ldk, void *stream) {
  if (!x || !cols_split || C <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return CRA5_ERR_ARG;
  if (ldk < C * kh * kw || (ldk % 32)) return CRA5_ERR_ARG;
  if (Ho != (H + 2 * ph - kh) / sh + 1 || Wo != (W + 2 * pw - kw) / sw + 1) return CRA5_ERR_ARG;
  const size_t total = (size_t)Ho * Wo * ldk;
  hipLaunchKernelGGL(conv_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, cols_split, C, H, W,
                     kh, kw, sh, sw, ph, pw, Ho, Wo, ldk);
  return (int)hipGetLastError();
}

int cra5_deconv_col2im_f32(const float *cols, const float *bias, float *out, int Cout, int Hi, int Wi, int kh, int kw,
                           int sh, int sw, int ph, int pw, int Ho, int Wo, int ldn, void *stream) {
  if (!cols || !out || Cout <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return CRA5_ERR_ARG;
  if (ldn < Cout * kh * kw || Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0) return CRA5_ERR_ARG;
  const size_t total = (size_t)Cout * Ho * Wo;
  hipLaunchKernelGGL(deconv_col2im_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, cols, bias, out, Cout,
                     Hi, Wi, kh, kw, sh, sw, ph, pw, Ho, Wo, ldn);
  return (int)hipGetLastError();
}

int cra5_unary_f32(const float *x, float *y, size_t n, int op, float slope, void *stream) {
  if (!x || !y || n == 0 || op < 0 || op > 1) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, op, slope);
  return (int)hipGetLastError();
}

int cra5_gaussian_conditional_f32(const float *y, const int32_t *sym_in, const float *scales, const float *means,
                                  const float *scale_table, int n_table, float scale_bound, float lik_bound,
                                  int32_t *idx, int32_t *sym, float *y_hat, float *lik, size_t n, void *stream) {
  if ((!y && !sym_in) || !scales || !means || n == 0) return CRA5_ERR_ARG;
  if (idx && (!scale_table || n_table < 1 || n_table > 256)) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(gaussian_conditional_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, sym_in,
                     scales, means, scale_table, idx ? n_table : 0, scale_bound, lik_bound, idx, sym, y_hat, lik, n);
  return (int)hipGetLastError();
}

int cra5_gaussian_conditional_compact_f32(const float *scales, const float *means, const float *scale_table, int n_table,
                                          float scale_bound, const int16_t *sym16_in, uint8_t *idx8, float *y_hat,
                                          size_t n, void *stream) {
  if (!means || n == 0 || (!idx8 && !y_hat)) return CRA5_ERR_ARG;
  if (idx8 && (!scales || !scale_table || n_table < 1 || n_table > 256)) return CRA5_ERR_ARG;
  if (y_hat && !sym16_in) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(gaussian_conditional_compact_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, scales,
                     means, scale_table, idx8 ? n_table : 0, scale_bound, sym16_in, idx8, y_hat, n);
  return (int)hipGetLastError();
}

int cra5_entropy_bottleneck_f32(const float *z, const int32_t *sym_in, const float *medians, const float *params,
                                float lik_bound, int32_t *sym, float *z_hat, float *lik, int C, int n_per_ch,
                                void *stream) {
  if ((!z && !sym_in) || !medians || C <= 0 || n_per_ch <= 0) return CRA5_ERR_ARG;
  if (lik && !params) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(entropy_bottleneck_kernel, dim3(grid_for((size_t)C * n_per_ch)), dim3(256), 0,
                     (hipStream_t)stream, z, sym_in, medians, params, lik_bound, sym, z_hat, lik, C, n_per_ch);
  return (int)hipGetLastError();
}

int cra5_rans_resolve_symbols_i32(const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs,
                                  int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets,
                                  uint32_t *start_range, uint32_t *raw, uint8_t *esc, void *stream) {
  if (!symbols || !indexes || !cdfs || !cdf_sizes || !offsets || !start_range || !raw || !esc) return CRA5_ERR_ARG;
  if (n == 0 || n_cdfs <= 0 || cdf_stride < 2) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(resolve_symbols_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, symbols, indexes, n,
                     cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets, start_range, raw, esc);
  return (int)hipGetLastError();
}

int cra5_rans_resolve_symbols_compact(const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs,
                                      int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets,
                                      uint32_t *start_range, uint16_t *rec16, int32_t *overflow, void *stream) {
  if (!symbols || !indexes || !cdfs || !cdf_sizes || !offsets || !start_range || !rec16 || !overflow) return CRA5_ERR_ARG;
  if (n == 0 || n_cdfs <= 0 || cdf_stride < 2) return CRA5_ERR_ARG;
  int rc = (int)hipMemsetAsync(overflow, 0, sizeof(int32_t), (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(resolve_symbols_compact_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, symbols,
                     indexes, n, cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets, start_range, rec16, overflow);
  return (int)hipGetLastError();
}

int cra5_gdn_f32(const float *x, const float *beta, const float *gamma, float *y, int B, int C, int HW, int inverse,
                 void *stream) {
  if (!x || !beta || !gamma || !y || B <= 0 || C <= 0 || HW <= 0 || C > 512) return CRA5_ERR_ARG;
  const int PIX = 64;
  dim3 grid((HW + PIX - 1) / PIX, B), block(256);
  hipLaunchKernelGGL(gdn_kernel, grid, block, (size_t)PIX * C * sizeof(float), (hipStream_t)stream, x, beta, gamma, y,
                     B, C, HW, inverse);
  return (int)hipGetLastError();
}

int cra5_event_create(void **ev) {
  hipEvent_t e;
  const int rc = (int)hipEventCreate(&e);
  *ev = (void *)e;
  return rc;
}
int cra5_event_record(void *ev, void *stream) { return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream); }
int cra5_event_elapsed_ms(void *start, void *stop, float *ms) {
  int rc = (int)hipEventSynchronize((hipEvent_t)stop);
  if (rc) return rc;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
int cra5_event_destroy(void *ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

int cra5_debug_range_counts(uint64_t *out2, int reset) {
#ifdef CRA5_RANGE_CHECK
  if (!out2) return CRA5_ERR_ARG;
  int rc = (int)hipDeviceSynchronize();
  if (rc) return rc;
  unsigned long long h[2] = {0, 0};
  if ((rc = cra5_range_counts_elementwise(h, reset))) return rc;
  if ((rc = cra5_range_counts_gemm(h, reset))) return rc;
  if ((rc = cra5_range_counts_attn(h, reset))) return rc;
  if ((rc = cra5_range_counts_attn_f32(h, reset))) return rc;
  if ((rc = cra5_range_counts_hyper(h, reset))) return rc;
  out2[0] = h[0];
  out2[1] = h[1];
  return rc;
#else
  (void)out2;
  (void)reset;
  return CRA5_ERR_UNAVAILABLE;
#endif
}

}  // extern "C"
