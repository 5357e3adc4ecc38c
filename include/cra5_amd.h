/* cra5_amd - MI355X (gfx950) native VAEformer encode/decode path: C ABI.
 *
 * This is the drop-in boundary.  Plain pointers and sizes
 * only; no torch / pybind types.  Every function returns 0 on success; device
 * launchers return the hipError_t of the launch (as int) otherwise, host functions
 * a negative cra5 status.  No function allocates device memory; the caller owns all
 * buffers and passes the hipStream_t to launch on (as void*).
 *
 * Reference interfaces each entry point replaces (paths relative to the reference
 * tree taohan10200/CRA5):
 *   - FFI #1 `compressai.ans`  cra5/models/compressai/cpp_exts/rans/rans_interface.cpp:361-381
 *   - FFI #2 `compressai._CXX` cra5/models/compressai/cpp_exts/ops/ops.cpp:111-118
 *   - the ATen op sites of the hot path (the reference has no device kernels of its
 *     own): cra5/models/vaeformer/vit_nlc.py and
 *     cra5/models/compressai/entropy_models/entropy_models.py, cited per function.
 */
#ifndef CRA5_AMD_H
#define CRA5_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (host functions) ------------------------------------------- */
#define CRA5_OK 0
#define CRA5_ERR_ALLOC (-1)
#define CRA5_ERR_INDEX (-2)     /* cdf index out of range                         */
#define CRA5_ERR_PMF_DOMAIN (-3)/* negative / non-finite pmf entry (ops.cpp:46-52) */
#define CRA5_ERR_PMF_ZERO (-4)  /* all-zero pmf (ops.cpp:60-64)                    */
#define CRA5_ERR_PMF_STEAL (-5) /* no bin can donate frequency                     */
#define CRA5_ERR_STREAM (-6)    /* truncated / corrupt rANS stream                 */
#define CRA5_ERR_ARG (-7)
#define CRA5_ERR_UNAVAILABLE (-8) /* entry point not compiled into this build flavour */
#define CRA5_ERR_RANGE (-9)     /* a value does not fit the compact record type: use the 32-bit entry point */
#define CRA5_ERR_DESYNC (-10)   /* every symbol decoded, but the coder is not back at its initial state / words are left
                                 * over: the indexes or tables are not the encoder's (or the stream is damaged) - the
                                 * decoded symbols are NOT the coded ones.  One-shot decoders only (the whole stream). */

int cra5_abi_version(void);

/* ============================ host: entropy coding ============================ */

/* RansEncoder.encode_with_indexes (rans_interface.cpp:202-213 -> :108-200).
 * `cdfs` is a dense int32 matrix [n_cdfs][cdf_stride] (the `_quantized_cdf`
 * buffer), `cdf_sizes` = `_cdf_length`, `offsets` = `_offset`.  The stream
 * (little-endian 32-bit words) is malloc'ed into *out; release with cra5_free.
 * Thread-safe: no global state, so many frames can be coded concurrently. */
int cra5_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t n,
                                  const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                  const int32_t *cdf_sizes, const int32_t *offsets,
                                  uint8_t **out, size_t *out_len);

/* The same encoder fed with symbols ALREADY resolved against their tables (the table
 * lookups move to the GPU, cra5_rans_resolve_symbols_i32 below, the host loop is pure state update).
 * start_range[i] = start | range << 16 of the bin (the escape bin for out-of-range symbols);
 * esc[i] = 0 for a regular symbol, 1 + number of 4-bit payload nibbles (1..9) for an escape whose
 * payload is raw[i] (rans_interface.cpp:121-150), 255 = invalid index (-> CRA5_ERR_INDEX).
 * Produces byte-for-byte the stream of cra5_rans_encode_with_indexes. */
int cra5_rans_encode_resolved(const uint32_t *start_range, const uint32_t *raw, const uint8_t *esc, size_t n,
                              uint8_t **out, size_t *out_len);
/* The same on COMPACT records (6 instead of 9 bytes per latent between device and host): rec16[i] = 0 for a regular
 * symbol, (1 + payload nibbles) << 12 | payload for an escape whose payload fits 12 bits; a record 0xFFFF (payload
 * beyond 12 bits / invalid index) returns CRA5_ERR_RANGE - encode from the 32-bit records then.  Same bytes. */
int cra5_rans_encode_resolved_compact(const uint32_t *start_range, const uint16_t *rec16, size_t n, uint8_t **out,
                                      size_t *out_len);

/* RansDecoder.decode_with_indexes (rans_interface.cpp:215-284). `out` has n slots.
 * Unlike the reference (which reads past the end of a corrupt stream) a truncated
 * stream returns CRA5_ERR_STREAM, and a stream that decodes to the end WITHOUT returning the coder to its initial
 * state (RANS64_L, every word consumed) returns CRA5_ERR_DESYNC: the reference hands back garbage silently there. */
int cra5_rans_decode_with_indexes(const uint8_t *encoded, size_t len, const int32_t *indexes,
                                  size_t n, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                  const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out);

/* The same decoder on COMPACT records (the frame path's device <-> host traffic: 3 bytes per latent instead of 8):
 * uint8 CDF indexes (the Gaussian tables have 64 rows), int16 symbols out.  A decoded symbol outside int16 returns
 * CRA5_ERR_RANGE with `out` partially written: decode again with cra5_rans_decode_with_indexes (same stream). */
int cra5_rans_decode_with_indexes_u8_i16(const uint8_t *encoded, size_t len, const uint8_t *indexes, size_t n,
                                         const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                         const int32_t *offsets, int16_t *out);

/* Code `n_streams` independent streams on a pool of `n_threads` host threads
 * (frames are independent units: entropy_models.py:263-272). Arrays of per-stream
 * pointers; tables may be shared between streams. rc[i] receives each status. */
int cra5_rans_encode_batch(int n_streams, const int32_t *const *symbols,
                           const int32_t *const *indexes, const size_t *n,
                           const int32_t *const *cdfs, const int *n_cdfs, const int *cdf_stride,
                           const int32_t *const *cdf_sizes, const int32_t *const *offsets,
                           uint8_t **out, size_t *out_len, int *rc, int n_threads);
int cra5_rans_decode_batch(int n_streams, const uint8_t *const *encoded, const size_t *len,
                           const int32_t *const *indexes, const size_t *n,
                           const int32_t *const *cdfs, const int *n_cdfs, const int *cdf_stride,
                           const int32_t *const *cdf_sizes, const int32_t *const *offsets,
                           int32_t *const *out, int *rc, int n_threads);

void cra5_free(void *p);

/* Stateful forms of the same coder (class decls rans_interface.hpp:49-113):
 *   BufferedRansEncoder: push() any number of (symbols, indexes, tables) groups, then flush()
 *   codes them all - last pushed symbol first - into ONE stream (rans_interface.cpp:108-200);
 *   RansDecoder.set_stream()/decode_stream(): the decoder keeps its state between calls
 *   (rans_interface.cpp:286-359).  Handles are not thread-safe; use one per thread. */
void *cra5_rans_encoder_new(void);
int cra5_rans_encoder_push(void *enc, const int32_t *symbols, const int32_t *indexes, size_t n,
                           const int32_t *cdfs, int n_cdfs, int cdf_stride,
                           const int32_t *cdf_sizes, const int32_t *offsets);
int cra5_rans_encoder_flush(void *enc, uint8_t **out, size_t *out_len);
void cra5_rans_encoder_free(void *enc);
void *cra5_rans_decoder_new(void);
int cra5_rans_decoder_set_stream(void *dec, const uint8_t *encoded, size_t len);
int cra5_rans_decoder_decode_stream(void *dec, const int32_t *indexes, size_t n, const int32_t *cdfs,
                                    int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                    const int32_t *offsets, int32_t *out);
void cra5_rans_decoder_free(void *dec);

/* pmf_to_quantized_cdf (ops.cpp:40-108). cdf has n+1 slots. */
int cra5_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf);

/* ============================ device: dense projections ======================= */

/* C[M,N] = epilogue( A[M,K] . W[N,K]^T ), fp32 in / fp32 MFMA accumulate
 * (v_mfma_f32_32x32x2_f32).  Replaces every nn.Linear / 1x1 Conv2d / im2col'ed
 * Conv2d / ConvTranspose2d GEMM on the path (vit_nlc.py:57-59,96,111,216-217,302,
 * 629-632,741; vaeformer.py:154-155).
 *   flags: CRA5_EPI_BIAS  -> + bias[n]
 *          CRA5_EPI_GELU  -> exact-erf GELU after bias (nn.GELU(), vit_nlc.py:52-69)
 *          CRA5_EPI_RES   -> + res[m*ldr + n] (residual stream / pos_embed; res may
 *                            alias C)
 * Requirements: K % 4 == 0, lda % 4 == 0, ldw % 4 == 0, 16-byte aligned A, W. */
#define CRA5_EPI_BIAS 1
#define CRA5_EPI_GELU 2
#define CRA5_EPI_RES 4
/* cra5_gemm_nt_split only: reduced-precision mode, hi.hi product only (plain f16 operands, fp32
 * accumulate, 1 MFMA per product instead of 3) - BASELINE.json configs[4], RMSE-gated.  Only the hi plane of
 * A / W is read, and only the hi plane of C_split is WRITTEN (its lo halves keep whatever they held: a
 * consumer of that matrix must run in this mode too). */
#define CRA5_GEMM_HI_ONLY 8
/* With CRA5_GEMM_HI_ONLY (big tiles, Kp % 64 == 0; CRA5_ERR_ARG otherwise): the operand / the split output is a PLAIN f16
 * matrix - a row is its k-values as contiguous halves (`lda_kp` / `ldw_kp` / `ldc_split_kp` then count HALVES per row:
 * a plain row may live in the first half of a split-layout row, pitch 2 * Kp).  One full 128-byte line per row and
 * 64-wide k-step instead of two half-lines: -13..-18 % per launch, bit-identical results. */
#define CRA5_GEMM_A_PLAIN 16
#define CRA5_GEMM_W_PLAIN 32
#define CRA5_GEMM_OUT_PLAIN 64
int cra5_gemm_nt_f32(const float *A, int lda, const float *W, int ldw, float *C, int ldc,
                     const float *bias, const float *res, int ldr, int M, int N, int K,
                     int flags, void *stream);

/* Same contraction, fp32-accurate, on the f16 matrix cores (3 x v_mfma_f32_32x32x16_f16 per
 * product, fp32 accumulate) with both operands in the "split-f16" layout: every fp32 value
 * x is stored as hi = f16(x), lo = f16(x - hi); a row is a sequence of 128-byte chunks of
 * [32 hi halves | 32 lo halves], Kp = K rounded up to a multiple of 32 with zero padding
 * (so a row has 2*Kp halves = the bytes of Kp floats).
 *   C = epi(wscale_inv * A . W^T); `wscale_inv` undoes the power-of-two scale applied to W
 *   when it was split.  Outputs: C (fp32, may be NULL) and/or C_split (split-f16 with row
 *   length ldc_split_kp, may be NULL) - e.g. fc1+GELU writes the split matrix fc2 reads.
 *   lda_kp / ldw_kp: row length (in K elements, multiples of 32) of the A / W buffers. */
int cra5_gemm_nt_split(const uint16_t *A, int lda_kp, const uint16_t *W, int ldw_kp, float *C,
                       int ldc, uint16_t *C_split, int ldc_split_kp, const float *bias,
                       const float *res, int ldr, int M, int N, int Kp, float wscale_inv,
                       int flags, void *stream);


/* fp32 [rows][K] (row stride ldx) * scale -> split-f16 [rows][2*Kp]. Used once per weight
 * tensor at load time and for the few activations no fused producer emits. */
int cra5_split_f16(const float *x, int ldx, uint16_t *out, int rows, int K, int Kp, float scale,
                   void *stream);

/* LayerNorm over the last dim (eps inside the sqrt), one row per wavefront
 * (partial(nn.LayerNorm, eps=1e-6): vit_nlc.py:266,278,381,626). D % 4 == 0, D <= 2048.
 * Outputs: y (fp32, may be NULL) and/or y_split (split-f16 rows of 2*split_kp halves, pad
 * columns zeroed, may be NULL).  split_plain != 0 (reduced-precision mode): y_split rows are PLAIN f16 - D
 * contiguous halves (+ zero pad to split_kp) at the start of each 2*split_kp-halves row - for a consumer
 * running with CRA5_GEMM_A_PLAIN. */
int cra5_layernorm_f32(const float *x, int ldx, const float *gamma, const float *beta, float *y,
                       int ldy, uint16_t *y_split, int split_kp, int rows, int D, float eps,
                       int split_plain, void *stream);

/* ============================ device: attention =============================== */

/* Streaming-softmax multi-head attention over windows of a (H, W) token grid, fp32
 * MFMA.  qkv: [H*W][3*C] rows = [q | k | v], each [heads][hd].  Windows are (wh, ww)
 * tiles of the grid zero-padded bottom/right to multiples of (wh, ww); a padded token
 * carries q = k = v = pad_row (the qkv bias) and attention is UNMASKED over the 576
 * tokens of a window (vit_nlc.py:219-258); wh = H, ww = W gives the global attention
 * of vit_nlc.py:94-112.  out: [H*W][C] (pre-projection), padded queries dropped.
 * hd must be 64 or 72.  out (fp32) and/or out_split (split-f16, rows of 2*split_kp halves;
 * pad columns are NOT written: zero the buffer once) may be NULL. */
int cra5_window_attention_f32(const float *qkv, const float *pad_row, float *out,
                              uint16_t *out_split, int split_kp, int C, int heads, int H, int W,
                              int wh, int ww, float scale, void *stream);

/* Same attention, fp32-accurate on the f16 matrix cores (hi/lo operand split, see
 * cra5_gemm_nt_split), reading the split-f16 qkv matrix [H*W][3C] the qkv projection wrote
 * (qkv_kp = 3C) and a split pad row, writing fp32 `out` and/or split-f16 `out_split`.
 * Requires head dim 64 and wh*ww % 32 == 0 (the 576-token windows and the 10 368-token global
 * attention); a window that is not the whole grid must have <= 1152 tokens (CRA5_ERR_ARG otherwise);
 * other shapes use cra5_window_attention_f32.  hi_only != 0: reduced-precision mode
 * (plain f16 q/k/v/p operands, 8 MFMAs per tile instead of 24; fp32 softmax statistics; only the hi plane of
 * out_split is written): 1 = split rows, 3 = PLAIN f16 rows (qkv, the pad row and out_split: element n at half n, row
 * pitches unchanged).  | CRA5_ATTN_PERSISTENT_UNITS: windows of >= 384 tokens run as persistent 12-wave work-groups
 * walking (window, head) units instead of 4-wave work-groups (an alternative schedule kept for measurements: slower on
 * MI355X; tokens of the key-split remainder units differ from the default schedule by fp32 rounding). */
#define CRA5_ATTN_PERSISTENT_UNITS 4
int cra5_window_attention_split(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split,
                                float *out, uint16_t *out_split, int out_kp, int C, int heads,
                                int H, int W, int wh, int ww, float scale, int hi_only,
                                void *stream);

/* The same call with a caller-owned device workspace.  With it, the whole-grid (global) launch runs the BALANCED
 * schedule (vit_nlc.py:94-112): every head owns CUs / heads work-group slots of 12 waves (3 per SIMD); a slot first runs
 * full passes of 12 query tiles over the whole key loop, then the remaining query tiles - grouped by 12 - are laid end to
 * end with their key loops and cut into one equal piece per slot, whose un-normalised partial softmaxes a merge kernel
 * combines in fixed order (deterministic): 324 + 222.75 key steps per CU instead of 2 x 324 on 256 CUs at 10 368 tokens
 * x 16 heads.  cra5_attention_balanced_plan: 1 when a plan exists for (n_tokens, heads) on the current device - the
 * bytes it needs go to *workspace_bytes (0 is possible: a plan without a key-split part; `workspace` may then be NULL) -
 * 0 when the shape has no balanced schedule (the call is then cra5_window_attention_split).  The plan depends on the
 * device's CU count, and tokens of the key-split part differ from the plain launch by fp32 rounding (same accuracy
 * class): outputs of g_a / g_s are reproducible per device and schedule, not across them - only the hyper-prior path
 * (its own kernels) has to be bit-stable between encoder and decoder.  cra5_attention_workspace_bytes: the bytes alone
 * (0 also when no plan exists - kept for callers that only size a buffer). */
int cra5_attention_balanced_plan(int n_tokens, int heads, size_t *workspace_bytes);
size_t cra5_attention_workspace_bytes(int n_tokens, int heads);
int cra5_window_attention_split_ws(const uint16_t *qkv_split, int qkv_kp, const uint16_t *pad_row_split,
                                   float *out, uint16_t *out_split, int out_kp, int C, int heads,
                                   int H, int W, int wh, int ww, float scale, int hi_only,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* Un-embed as ONE fused launch pair (vit_nlc.py:628-630, 666-669: ConvTranspose2d(D -> C, kernel (11, 10), stride
 * (10, 10)) on the token grid): x[C][H][W] = overlap-add(A[M = Hp*Wp tokens][K] . W[N = C*110][K]^T) (* std + mean per
 * channel when both are given - the de-normalisation of cra5_api.py:268-271 fused as in cra5_col2im_f32).  The GEMM's
 * epilogue stores every accumulator straight into the image (the nine rows of a patch row that have one contribution)
 * or into `side` (the ky = 0 / ky = 10 rows: [C][Hp][2][W] floats, cra5_unembed_side_bytes), and a small second kernel
 * adds the overlap pairs in fixed order: no [tokens][C*110] column matrix, no pass over it, deterministic, and
 * bit-identical to cra5_gemm_nt_split + cra5_col2im_f32.  Only this geometry (kh 11, kw 10, strides 10, W % 4 == 0,
 * Kp <= 8192); cra5_unembed_side_bytes returns 0 and the launcher CRA5_ERR_ARG for anything else - use the two-call
 * form then.  hi_only: the reduced-precision mode of cra5_gemm_nt_split - bit 0 (1) hi-only products, | 2 the A rows are
 * plain f16 rows, | 4 the weight is a packed plain [N][Kp] f16 matrix (bits 1 / 2 only with bit 0; see CRA5_GEMM_A_PLAIN). */
size_t cra5_unembed_side_bytes(int C, int H, int W, int kh, int kw, int sh, int sw);
int cra5_gemm_nt_split_unembed(const uint16_t *A, int lda_kp, const uint16_t *W_split, int ldw_kp, float *x,
                               float *side, size_t side_bytes, const float *mean, const float *stdv, int M, int Kp,
                               float wscale_inv, int C, int H, int W, int kh, int kw, int sh, int sw, int hi_only,
                               void *stream);

/* ============================ device: layout / conv edges ===================== */

/* Patch gather for a strided Conv2d as GEMM (vit_nlc.py:302-308), fused with the API's
 * normalisation (x - mean[c]) / std[c] (cra5_api.py:264-266) when mean != NULL.
 * x: [C][H][W]; cols: [Hp*Wp][ldk], column (c*kh + i)*kw + j; columns >= C*kh*kw are
 * left untouched (keep them zero).  cols (fp32) and/or cols_split (split-f16, Kp = ldk,
 * ldk % 32 == 0) may be NULL.  split_plain != 0 (reduced-precision mode): cols_split rows are PLAIN f16 - C*kh*kw
 * contiguous halves, zero padded to ldk, at the start of each 2*ldk-halves row (ERA5 patch geometry, or ldk == C*kh*kw). */
int cra5_im2col_f32(const float *x, const float *mean, const float *std, float *cols,
                    uint16_t *cols_split, int C, int H, int W, int kh, int kw, int sh, int sw,
                    int Hp, int Wp, int ldk, int split_plain, void *stream);

/* Overlap-add scatter of a ConvTranspose2d computed as GEMM (vit_nlc.py:628-630,
 * 666-669), fused with de-normalisation x*std[c] + mean[c] (cra5_api.py:268-271) when
 * mean != NULL.  cols: [Hp*Wp][ldn], column (c*kh + i)*kw + j;  x: [C][H][W] with
 * H = (Hp-1)*sh + kh, W = (Wp-1)*sw + kw. */
int cra5_col2im_f32(const float *cols, const float *mean, const float *std, float *x, int C,
                    int H, int W, int kh, int kw, int sh, int sw, int Hp, int Wp, int ldn,
                    void *stream);

/* Finiteness probe of the range guard: partials[b] = sum of x[i * stride] over block b's share of i = 0 .. ceil(n / stride)
 * - 1, b = 0 .. CRA5_PROBE_PARTIALS - 1 (written, never accumulated: no memset, deterministic).  A partial is non-finite as
 * soon as one addend is; the caller copies the partials to the host with the phase's other results and tests them there. */
#define CRA5_PROBE_PARTIALS 256
int cra5_probe_sums_f32(const float *x, size_t n, size_t stride, float *partials, void *stream);

/* out[c][r] = in[r][c] (token-major <-> NCHW plumbing, vit_nlc.py:484, 684). */
int cra5_transpose_f32(const float *in, int ld_in, float *out, int ld_out, int rows, int cols,
                       void *stream);

/* 'b h w (p1 p2 c) -> b c (h p1) (w p2)' (vit_nlc.py:672-679): lin [Hz*Wz][p1*p2*Cout]
 * -> out [Cout][Hz*p1][Wz*p2]. */
int cra5_pixel_shuffle_f32(const float *lin, float *out, int Hz, int Wz, int p1, int p2,
                           int Cout, void *stream);

/* CNN zoo (bmshj2018-factorized / -hyperprior, mbt2018-mean: cra5/models/compressai/models/google.py:64-508;
 * conv / deconv of models/utils.py:128-147).  Conv2d(k, s, padding p): zero-padded patch gather into the
 * split-f16 GEMM operand, cols_split [Ho*Wo][2*ldk], column (c*kh + i)*kw + j, pad columns zero.
 * ConvTranspose2d(k, s, padding p, output_padding): cols = in^T . W (GEMM, fp32 [Hi*Wi][ldn], column
 * (co*kh + i)*kw + j), then out[co][y][x] = bias[co] + sum of the contributions with y = yi*s - p + i. */
int cra5_conv_im2col_f32(const float *x, uint16_t *cols_split, int C, int H, int W, int kh, int kw, int sh, int sw,
                         int ph, int pw, int Ho, int Wo, int ldk, void *stream);
int cra5_deconv_col2im_f32(const float *cols, const float *bias, float *out, int Cout, int Hi, int Wi, int kh, int kw,
                           int sh, int sw, int ph, int pw, int Ho, int Wo, int ldn, void *stream);
/* y = relu(x) / leaky_relu(x, slope) (op 0; slope 0 = ReLU) or |x| (op 1); x == y allowed. */
int cra5_unary_f32(const float *x, float *y, size_t n, int op, float slope, void *stream);

/* ============================ device: entropy models ========================== */

/* GaussianConditional, eval mode (entropy_models.py:645-685, 155-201):
 *   s = max(scale, bound);  idx = (n_table-1) - #{t in table[0:n_table-1] : s <= t}
 *   sym = (int)rintf(y - mean);  y_hat = sym + mean
 *   lik = max(Phi((.5-|y_hat-mean|)/s) - Phi((-.5-|y_hat-mean|)/s), lik_bound)
 * Any output pointer may be NULL.  If y == NULL and sym_in != NULL the kernel
 * de-quantises instead: y_hat = sym_in + mean (decode side, :192-201). */
int cra5_gaussian_conditional_f32(const float *y, const int32_t *sym_in, const float *scales,
                                  const float *means, const float *scale_table, int n_table,
                                  float scale_bound, float lik_bound, int32_t *idx, int32_t *sym,
                                  float *y_hat, float *lik, size_t n, void *stream);
/* Decode-side halves of the same op on compact records: idx8 = the CDF index as uint8 (n_table <= 256), or - with
 * sym16_in - y_hat = sym16_in + mean.  Same arithmetic as cra5_gaussian_conditional_f32 (bit-identical idx / y_hat). */
int cra5_gaussian_conditional_compact_f32(const float *scales, const float *means, const float *scale_table, int n_table,
                                          float scale_bound, const int16_t *sym16_in, uint8_t *idx8, float *y_hat,
                                          size_t n, void *stream);

/* EntropyBottleneck, eval mode (entropy_models.py:434-510, 529-542): z is [C][n_per_ch];
 *   sym = (int)rintf(z - median[c]);  z_hat = sym + median[c]
 *   lik = max(sigmoid(logits(z_hat+.5)) - sigmoid(logits(z_hat-.5)), lik_bound)
 * params: per channel 58 floats: softplus(M0)[3] b0[3] tanh(f0)[3] | softplus(M1)[9] b1[3]
 * tanh(f1)[3] | M2.. | M3.. | softplus(M4)[3] b4[1]  (prepared on the host).
 * If z == NULL and sym_in != NULL: z_hat = sym_in + median (decode side). */
int cra5_entropy_bottleneck_f32(const float *z, const int32_t *sym_in, const float *medians,
                                const float *params, float lik_bound, int32_t *sym, float *z_hat,
                                float *lik, int C, int n_per_ch, void *stream);

/* GDN / IGDN (cra5/models/compressai/layers/gdn.py:76-92): x, y: [B][C][HW];
 * y = x * rsqrt(beta[i] + sum_j gamma[i][j] x_j^2)   (inverse: * sqrt).
 * beta / gamma are the re-parametrised (effective) values. */
/* Device side of cra5_rans_encode_resolved: symbols / indexes / tables are DEVICE pointers (the
 * model's _quantized_cdf [n_cdfs][cdf_stride], _cdf_length, _offset buffers); writes start_range,
 * raw and esc (n entries each) as described there.  rans_interface.cpp:121-150 per element. */
int cra5_rans_resolve_symbols_i32(const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs,
                                  int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets,
                                  uint32_t *start_range, uint32_t *raw, uint8_t *esc, void *stream);
/* Device side of cra5_rans_encode_resolved_compact; *overflow (device int32) is zeroed on the stream, then set to 1 when
 * any record is 0xFFFF. */
int cra5_rans_resolve_symbols_compact(const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs,
                                      int n_cdfs, int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets,
                                      uint32_t *start_range, uint16_t *rec16, int32_t *overflow, void *stream);

int cra5_gdn_f32(const float *x, const float *beta, const float *gamma, float *y, int B, int C,
                 int HW, int inverse, void *stream);

/* ---- hyper-prior path (HyperpriorEncoder / HyperpriorDecoder, vit_nlc.py:488-551, 696-748) --------
 * Latency-bound sizes (648 tokens x 360 channels): csrc/hyper.hip.  Both are deterministic (fixed
 * reduction order) - h_s runs on the encode and on the decode side and must agree bit for bit. */

/* cra5_gemm_nt_split for small M: one wave per 32 x 32..128 tile, operands straight from L2, optional
 * in-block split-K.  Same arguments; C_split pad columns (N .. ldc_split_kp) are written as zeros.
 * ps_Hz > 0 selects the fused un-embed store of HyperpriorDecoder (vit_nlc.py:672-679,
 * `b h w (p1 p2 c) -> b c (h p1) (w p2)`): M = ps_Hz * ps_Wz tokens, the N = Cout * ps_p1 * ps_p2 weight
 * rows must be in (c, p1, p2) order (ps_p2 must be 4), and C is the image [Cout][ps_Hz*ps_p1][ps_Wz*4]. */
int cra5_small_gemm_nt_split(const uint16_t *A, int lda_kp, const uint16_t *W, int ldw_kp, float *C, int ldc,
                             uint16_t *C_split, int ldc_split_kp, const float *bias, const float *res, int ldr,
                             int M, int N, int Kp, float wscale_inv, int flags, int ps_Hz, int ps_Wz, int ps_p1,
                             int ps_p2, void *stream);

/* Attention.forward (vit_nlc.py:94-112) over all n_tok tokens, exact fp32 MFMA, head dim 72 | 64:
 * qkv fp32 [n_tok][3C] -> out fp32 [n_tok][C] and / or out_split (pad columns must already be zero). */
int cra5_hyper_attention_f32(const float *qkv, float *out, uint16_t *out_split, int split_kp, int n_tok, int C,
                             int heads, float scale, void *stream);

/* Device-side timing helpers for bench.py (HIP events on the launch stream). */
int cra5_event_create(void **ev);
int cra5_event_record(void *ev, void *stream);
int cra5_event_elapsed_ms(void *start, void *stop, float *ms);
int cra5_event_destroy(void *ev);

/* Host <-> device frame copies through a caller-owned PINNED staging buffer of >= `bytes` (csrc/runtime.hip): the copy
 * is cut into `chunk_bytes` chunks; `n_threads` host threads (the caller is one of them) memcpy chunk c + 1 between
 * pageable and pinned memory while the DMA engine moves chunk c on `stream`.  Replaces the `.to(device)` / `.cpu()` of
 * a 1.11 GB frame in cra5_api.py:81-125,153-192.  h2d: returns once the DMA engine has read the last chunk out
 * of `pinned` (the staging buffer and `dst_dev` are the caller's again: the next frame may re-use both); the team is
 * capped at the CPUs the process may run on and its wait loops yield; d2h: returns when `dst_host` holds
 * all bytes (the copies are queued behind whatever `stream` already holds). */
int cra5_copy_h2d_staged(void *dst_dev, const void *src_host, void *pinned, size_t bytes, size_t chunk_bytes,
                         int n_threads, void *stream);
int cra5_copy_d2h_staged(void *dst_host, const void *src_dev, void *pinned, size_t bytes, size_t chunk_bytes,
                         int n_threads, void *stream);

/* Shader-clock telemetry for bench.py (csrc/runtime.hip).  cra5_clock_probe: ONE wave that stays for `window_ticks`
 * of the 100 MHz wall clock (<= 10 ms) and stores {wall0, cycles0, wall1, cycles1} into slot4_dev[0..3]: effective
 * shader clock of the window = (cycles1 - cycles0) / (wall1 - wall0) x 100 MHz, on the CU the wave landed on, beside
 * whatever else runs.  Short probes on purpose: a resident sampler wave would block every stream that shares its
 * hardware queue.  cra5_clock_stamp writes the wall clock into one device word in stream order (brackets queued work). */
int cra5_clock_probe(uint64_t *slot4_dev, int window_ticks, void *stream);
int cra5_clock_stamp(uint64_t *slot_dev, void *stream);

/* Range audit of the split-f16 producers (csrc/split.h): out[0] = elements with |x| >= 65504 (clipped
 * by the saturating split), out[1] = non-finite elements, counted since the last reset by every
 * LayerNorm / GEMM-epilogue / attention / patch-gather store.  Only in the `rangecheck` build flavour
 * (python -m cra5_amd.build --flavour rangecheck, -DCRA5_RANGE_CHECK); CRA5_ERR_UNAVAILABLE otherwise.
 * Synchronises the device. */
int cra5_debug_range_counts(uint64_t *out2, int reset);

#ifdef __cplusplus
}
#endif
#endif /* CRA5_AMD_H */
