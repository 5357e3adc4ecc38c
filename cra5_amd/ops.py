"""Thin Python launchers over the C ABI.  torch tensors are containers only: every
function passes raw `data_ptr()`s + sizes + the current HIP stream to the native
library.  Device entry points refuse non-GPU tensors (no CPU fallback)."""
import contextlib
import ctypes
import os
import threading

import numpy as np
import torch

from ._lib import check, lib

EPI_BIAS, EPI_GELU, EPI_RES, GEMM_HI_ONLY = 1, 2, 4, 8


_TLS = threading.local()


def _stream():
    s = getattr(_TLS, "stream", None)
    if s is not None:
        return s
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@contextlib.contextmanager
def stream_scope():
    """Inside the scope this thread's launches go to the stream that is torch's current one AT ENTRY, looked up once
    (torch.cuda.current_stream().cuda_stream costs 2.7 us per launch - 30 % of a launch's host time; a frame's GPU phase
    is ~150 launches on one stream).  Do not switch torch streams inside the scope."""
    prev = getattr(_TLS, "stream", None)
    _TLS.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        yield
    finally:
        _TLS.stream = prev


class ClockSampler:
    """Shader-clock telemetry (cra5_clock_probe): a host thread launches a one-wave probe of `window_us` every
    `period_ms` on its own stream while the workload runs; `summary()` gives the effective shader clock the probes saw.
    bench.py runs it over the timed region so that two JSON lines from two boxes can be told apart by the clock they
    sustained.  (Short probes, not a resident wave: a long kernel blocks the streams sharing its hardware queue.)"""

    def __init__(self, device, n_max=4096, period_ms=5.0, window_us=50):
        self.n_max, self.period, self.window = int(n_max), float(period_ms) * 1e-3, int(window_us * 100)
        self.buf = torch.zeros(4 * self.n_max, dtype=torch.int64, device=device)
        self.stream = torch.cuda.Stream(device=device)
        self.n, self._on, self._th = 0, False, None

    def probe(self):
        """one probe on the sampler's stream (callers that pace the probes themselves)"""
        if self.n >= self.n_max:
            return
        check(lib().cra5_clock_probe(ctypes.c_void_p(self.buf.data_ptr() + 32 * self.n), self.window,
                                     ctypes.c_void_p(self.stream.cuda_stream)), "cra5_clock_probe")
        self.n += 1

    def start(self):
        import time
        self.n, self._on = 0, True

        def run():
            while self._on and self.n < self.n_max:
                self.probe()
                time.sleep(self.period)
        self._th = threading.Thread(target=run, daemon=True, name="cra5-clock")
        self._th.start()

    def stop(self):
        self._on = False
        if self._th is not None:
            self._th.join()
            self._th = None
        self.stream.synchronize()

    def summary(self, w0=None, w1=None):
        """w0 / w1: keep the probes that lie inside this wall-clock window (cra5_clock_stamp values)."""
        import numpy as np
        if self.n < 1:
            return None
        b = self.buf.cpu().numpy()[: 4 * self.n].reshape(self.n, 4).astype(np.int64)
        ok = b[:, 2] > b[:, 0]
        if w0 is not None:
            ok &= (b[:, 0] >= w0) & (b[:, 2] <= w1)
        b = b[ok]
        if len(b) < 1:
            return None
        ghz = (b[:, 3] - b[:, 1]) / ((b[:, 2] - b[:, 0]) * 10.0)
        return {"shader_ghz_mean": float(ghz.mean()), "shader_ghz_p10": float(np.percentile(ghz, 10)),
                "shader_ghz_median": float(np.median(ghz)), "shader_ghz_p90": float(np.percentile(ghz, 90)),
                "probes": int(len(b)), "probe_window_us": self.window / 100.0,
                "span_ms": float((b[-1, 2] - b[0, 0]) / 1e5)}


class KernelTimer:
    """Per-launch device timing with HIP events recorded on the launch stream
    (cra5_event_* in the C ABI).  Used by bench.py for the `roofline` object: it brackets
    every launch of one kernel family, so sum(work) / sum(duration) is that kernel's
    achieved rate over the timed region."""

    def __init__(self, sample_every=1):
        """(thread-safe: frame threads record concurrently)
        sample_every = n: bracket every n-th launch only (per calling thread).  Event records
        are queue packets between the kernels; at ~400 timed launches per frame they cost ~5 %
        of the frame rate, sampling keeps the timed region honest."""
        self.records = {}   # kind -> list of (start_ev, stop_ev, work)
        self._free = []
        self._lock = threading.Lock()
        self.sample_every = max(1, int(sample_every))
        self._tls = threading.local()

    def _ev(self):
        with self._lock:
            if self._free:
                return self._free.pop()
        e = ctypes.c_void_p()
        check(lib().cra5_event_create(ctypes.byref(e)), "cra5_event_create")
        return e

    def start(self):
        if self.sample_every > 1:
            n = getattr(self._tls, "n", 0) + 1
            self._tls.n = n
            if n % self.sample_every:
                return None
        e = self._ev()
        check(lib().cra5_event_record(e, _stream()), "cra5_event_record")
        return e

    def stop(self, kind, start_ev, work):
        if start_ev is None:
            return
        e = self._ev()
        check(lib().cra5_event_record(e, _stream()), "cra5_event_record")
        with self._lock:
            self.records.setdefault(kind, []).append((start_ev, e, work))

    def summary(self):
        """kind -> dict(launches, work, ms). Synchronises on the recorded events."""
        out = {}
        with self._lock:
            records, self.records = self.records, {}
        for kind, recs in records.items():
            ms_tot, work, done = 0.0, 0.0, []
            for s, e, w in recs:
                ms = ctypes.c_float()
                check(lib().cra5_event_elapsed_ms(s, e, ctypes.byref(ms)), "cra5_event_elapsed_ms")
                ms_tot += ms.value
                work += w
                done += [s, e]
            with self._lock:       # frame threads pop from _free concurrently (_ev)
                self._free += done
            out[kind] = dict(launches=len(recs), work=work, ms=ms_tot)
        return out


TIMER = None  # set to a KernelTimer by bench.py

COPY_CHUNK = 32 << 20
COPY_THREADS = 8     # host threads of one staged copy; cra5_api / VAEformer pass RuntimeConfig.copy_threads explicitly


def copy_h2d_staged(dst, src_np, pinned, threads=None):
    """host numpy array (C-contiguous) -> device tensor `dst` of the same byte size, through the pinned tensor
    `pinned`, on the current stream (cra5_copy_h2d_staged): chunked, host memcpy overlapped with the DMA."""
    n = src_np.nbytes
    if not (src_np.flags["C_CONTIGUOUS"] and dst.is_cuda and dst.is_contiguous() and dst.numel() * dst.element_size() == n):
        raise ValueError("copy_h2d_staged: the host array must be C-contiguous and the device tensor contiguous, of the same byte size")
    if not (pinned.is_pinned() and pinned.numel() * pinned.element_size() >= n):
        raise ValueError("copy_h2d_staged: `pinned` must be a pinned tensor of at least the frame's byte size")
    check(lib().cra5_copy_h2d_staged(_p(dst), ctypes.c_void_p(src_np.ctypes.data), ctypes.c_void_p(pinned.data_ptr()),
                                     n, COPY_CHUNK, threads or COPY_THREADS, _stream()), "cra5_copy_h2d_staged")
    return dst


def copy_d2h_staged(dst_np, src, pinned, threads=None):
    """device tensor -> host numpy array through `pinned`; returns when `dst_np` holds the data."""
    n = dst_np.nbytes
    if not (dst_np.flags["C_CONTIGUOUS"] and dst_np.flags["WRITEABLE"] and src.is_cuda and src.is_contiguous()
            and src.numel() * src.element_size() == n):
        raise ValueError("copy_d2h_staged: the host array must be writeable and C-contiguous, the device tensor contiguous, "
                         "of the same byte size")
    if not (pinned.is_pinned() and pinned.numel() * pinned.element_size() >= n):
        raise ValueError("copy_d2h_staged: `pinned` must be a pinned tensor of at least the frame's byte size")
    check(lib().cra5_copy_d2h_staged(ctypes.c_void_p(dst_np.ctypes.data), _p(src), ctypes.c_void_p(pinned.data_ptr()),
                                     n, COPY_CHUNK, threads or COPY_THREADS, _stream()), "cra5_copy_d2h_staged")
    return dst_np


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cra5_amd device op called with a non-GPU tensor: the HIP path is the only path "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        if t.dtype not in (torch.float32, torch.int32, torch.int16):
            raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    # (a plain int: every entry point declares its argtypes, ctypes converts int -> void * itself)
    return None if t is None else t.data_ptr()


def _row_stride(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "expected a 2-D tensor with unit inner stride"
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm_nt(a, w, bias=None, res=None, gelu=False, out=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T).  a / w / res / out may be row-strided views."""
    _dev(a, w, bias, res, out)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RES if res is not None else 0)
    ev = TIMER.start() if TIMER is not None else None
    check(lib().cra5_gemm_nt_f32(_p(a), _row_stride(a), _p(w), _row_stride(w), _p(out), _row_stride(out), _p(bias),
                                 _p(res), _row_stride(res) if res is not None else 0, M, N, K, flags, _stream()),
          "cra5_gemm_nt_f32")
    if ev is not None:
        TIMER.stop("gemm_nt_f32", ev, 2.0 * M * N * K)
    return out


class SplitMat:
    """A [rows, K] fp32 matrix in the split-f16 layout of csrc/gemm_split_f16.hip: `data` is a
    uint16 tensor [rows, 2*Kp] (128-byte chunks of 32 hi | 32 lo halves), Kp = K rounded up to
    32 (zero padded), value = (hi + lo) * scale_inv."""

    __slots__ = ("data", "rows", "K", "Kp", "scale_inv", "plain")

    def __init__(self, data, rows, K, Kp, scale_inv=1.0, plain=False):
        self.data, self.rows, self.K, self.Kp, self.scale_inv = data, rows, K, Kp, scale_inv
        # plain = True (reduced-precision mode, round 5): the rows hold PLAIN f16 - element n at half n, no lo plane; the
        # row pitch is data.shape[1] halves (2 * Kp when the plain row sits in the first half of a split-layout buffer, Kp
        # for a plain weight copy).  Set by the producer of each call.
        self.plain = plain

    @property
    def pitch(self):
        """what the C ABI takes as lda_kp / ldw_kp / ldc_split_kp: k-elements per split row, halves per plain row"""
        return self.data.shape[1] if self.plain else self.Kp

    def plain_copy(self):
        """A PLAIN f16 copy of this (weight) matrix: its hi plane, [rows, Kp] contiguous, same scale."""
        assert not self.plain
        hi = self.data.view(self.rows, self.Kp // 32, 2, 32)[:, :, 0].reshape(self.rows, self.Kp).contiguous()
        return SplitMat(hi, self.rows, self.K, self.Kp, self.scale_inv, plain=True)

    @staticmethod
    def empty(rows, K, device, zero=False):
        Kp = (K + 31) // 32 * 32
        # int16 storage (torch has no uint16 arithmetic; only the bytes matter)
        data = (torch.zeros if zero else torch.empty)((rows, 2 * Kp), device=device, dtype=torch.int16)
        return SplitMat(data, rows, K, Kp)

    def to_float(self):
        """Reconstruct the fp32 values (tests / debugging)."""
        if self.plain:
            return self.data.view(torch.float16)[:, : self.K].float() * self.scale_inv
        h = self.data.view(torch.float16).view(self.rows, self.Kp // 32, 2, 32).float()
        return ((h[:, :, 0] + h[:, :, 1]).reshape(self.rows, self.Kp)[:, : self.K]) * self.scale_inv


def _devs(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cra5_amd device op called with a non-GPU tensor: the HIP path is the only path")


def split_f16(x, scale_pow2=None, out=None):
    """fp32 [rows, K] (row-strided ok) -> SplitMat.  scale_pow2: None = 1.0; 'auto' = the
    power of two that brings max|x| into [2^12, 2^13) (weights)."""
    _dev(x)
    rows, K = x.shape
    scale = 1.0
    if scale_pow2 == "auto":
        amax = float(x.abs().max())
        if amax > 0 and amax == amax and amax != float("inf"):
            import math
            scale = 2.0 ** (12 - math.floor(math.log2(amax)))
    elif scale_pow2 is not None:
        scale = float(scale_pow2)
    if out is None:
        out = SplitMat.empty(rows, K, x.device)
    check(lib().cra5_split_f16(_p(x), _row_stride(x), _p(out.data), rows, K, out.Kp, scale, _stream()),
          "cra5_split_f16")
    out.scale_inv = 1.0 / scale
    out.plain = False
    return out


GEMM_A_PLAIN, GEMM_W_PLAIN, GEMM_OUT_PLAIN = 16, 32, 64


def plain_ok(M, N, Kp):
    """Shapes on which cra5_gemm_nt_split takes plain-f16 operands / writes a plain output (the wide reduced-precision
    form: big tiles, 64-wide k-steps, no long-K chain through a split output)."""
    return ((M + 127) // 128) * ((N + 127) // 128) >= 256 and Kp % 64 == 0


def gemm_nt_split(a, w, bias=None, res=None, gelu=False, out=None, out_split=None, want_f32=True, hi_only=False,
                  out_plain=False):
    """epi(a @ w^T) with a, w SplitMat (same K).  out: fp32 [M, N] (row-strided ok) unless
    want_f32=False; out_split: SplitMat [M, N] to receive the split-f16 result.  Reduced-precision mode (hi_only): a / w
    may be PLAIN matrices (SplitMat.plain) and out_plain=True writes out_split's rows plain."""
    _devs(a.data, w.data)
    _dev(bias, res, out)
    M, N = a.rows, w.rows
    assert a.Kp == w.Kp and a.K == w.K, (a.K, w.K)
    assert a.scale_inv == 1.0, "activations are split unscaled"
    if out is None and want_f32:
        out = torch.empty((M, N), device=a.data.device, dtype=torch.float32)
    if out_split is not None:
        assert out_split.rows == M and out_split.K == N
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RES if res is not None else 0)
    if hi_only:
        flags |= GEMM_HI_ONLY
    if a.plain or w.plain or out_plain:
        assert hi_only, "plain-f16 operands exist in the reduced-precision mode only"
        flags |= (GEMM_A_PLAIN if a.plain else 0) | (GEMM_W_PLAIN if w.plain else 0) | (GEMM_OUT_PLAIN if out_plain else 0)
    if out_split is not None:
        out_split.plain = bool(out_plain)
    ev = TIMER.start() if TIMER is not None else None
    check(lib().cra5_gemm_nt_split(_p(a.data), a.pitch, _p(w.data), w.pitch, _p(out),
                                   _row_stride(out) if out is not None else 0,
                                   _p(out_split.data) if out_split is not None else None,
                                   out_split.pitch if out_split is not None else 0, _p(bias), _p(res),
                                   _row_stride(res) if res is not None else 0, M, N, a.Kp, float(w.scale_inv),
                                   flags, _stream()), "cra5_gemm_nt_split")
    if ev is not None:
        # same rule as gemm_dispatch(): < 256 128x128 tiles -> the 64x64-tile instantiation (hyper-prior
        # and head GEMMs: microseconds, launch-bound); everything else is the 192/256-row-tile kernel
        small = ((M + 127) // 128) * ((N + 127) // 128) < 256
        TIMER.stop("gemm_nt_split_small" if small else "gemm_nt_split", ev, 2.0 * M * N * a.K)
    return out


def unembed_side_bytes(C, H, W, kh, kw, sh, sw):
    """Bytes of the side buffer of gemm_unembed for this geometry; 0: the fused form does not take it."""
    return int(lib().cra5_unembed_side_bytes(C, H, W, kh, kw, sh, sw))


def gemm_unembed(a, w, C, H, W, kh, kw, sh, sw, side, mean=None, std=None, out=None, hi_only=False):
    """Fused un-embed (cra5_gemm_nt_split_unembed): a SplitMat [Hp*Wp, K] (tokens), w SplitMat [C*kh*kw, K] -> the image
    x [C, H, W] (de-normalised when mean / std are given).  `side`: device float buffer of >= unembed_side_bytes()."""
    _devs(a.data, w.data)
    _dev(side, mean, std, out)
    assert a.Kp == w.Kp and a.K == w.K and w.rows == C * kh * kw and a.scale_inv == 1.0
    if out is None:
        out = torch.empty((C, H, W), device=a.data.device, dtype=torch.float32)
    assert out.is_contiguous() and tuple(out.shape) == (C, H, W) and side.is_contiguous()
    ev = TIMER.start() if TIMER is not None else None
    hi = (1 if hi_only else 0) | (2 if a.plain else 0) | (4 if w.plain else 0)
    assert hi in (0, 1) or hi_only, "plain-f16 operands exist in the reduced-precision mode only"
    check(lib().cra5_gemm_nt_split_unembed(_p(a.data), a.pitch, _p(w.data), w.pitch, _p(out), _p(side),
                                           side.numel() * side.element_size(), _p(mean), _p(std), a.rows, a.Kp,
                                           float(w.scale_inv), C, H, W, kh, kw, sh, sw, hi, _stream()),
          "cra5_gemm_nt_split_unembed")
    if ev is not None:
        TIMER.stop("gemm_nt_split", ev, 2.0 * a.rows * w.rows * a.K)
    return out


def small_gemm_nt_split(a, w, bias=None, res=None, gelu=False, out=None, out_split=None, want_f32=True,
                        unembed=None):
    """gemm_nt_split for the hyper-prior sizes (csrc/hyper.hip).  unembed = (Hz, Wz, p1, p2): `out` is the
    image [N / (p1*p2), Hz*p1, Wz*p2] and w's rows are in (c, p1, p2) order (HyperpriorDecoder un-embed)."""
    _devs(a.data, w.data)
    _dev(bias, res, out)
    M, N = a.rows, w.rows
    assert a.Kp == w.Kp and a.K == w.K, (a.K, w.K)
    assert a.scale_inv == 1.0, "activations are split unscaled"
    if unembed is not None:
        Hz, Wz, p1, p2 = unembed
        assert out is not None and out.is_contiguous() and out.numel() == M * N
    else:
        Hz = Wz = p1 = p2 = 0
        if out is None and want_f32:
            out = torch.empty((M, N), device=a.data.device, dtype=torch.float32)
    if out_split is not None:
        assert out_split.rows == M and out_split.K == N
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_GELU if gelu else 0) | (EPI_RES if res is not None else 0)
    ev = TIMER.start() if TIMER is not None else None
    if out_split is not None:
        out_split.plain = False      # (split rows: workspace matrices change layout with the mode)
    check(lib().cra5_small_gemm_nt_split(_p(a.data), a.Kp, _p(w.data), w.Kp, _p(out),
                                         _row_stride(out) if (out is not None and unembed is None) else 0,
                                         _p(out_split.data) if out_split is not None else None,
                                         out_split.Kp if out_split is not None else 0, _p(bias), _p(res),
                                         _row_stride(res) if res is not None else 0, M, N, a.Kp, float(w.scale_inv),
                                         flags, Hz, Wz, p1, p2, _stream()), "cra5_small_gemm_nt_split")
    if ev is not None:
        TIMER.stop("gemm_nt_split_small", ev, 2.0 * M * N * a.K)
    return out


def hyper_attention(qkv, heads, out=None, out_split=None, want_f32=True):
    """Global attention of the hyper-prior blocks: qkv fp32 [n, 3C] -> [n, C] (exact fp32 MFMA)."""
    _dev(qkv, out)
    n, C3 = qkv.shape
    C = C3 // 3
    assert qkv.is_contiguous()
    if out is None and want_f32:
        out = torch.empty((n, C), device=qkv.device, dtype=torch.float32)
    if out_split is not None:
        assert out_split.rows == n and out_split.K == C
    ev = TIMER.start() if TIMER is not None else None
    if out_split is not None:
        out_split.plain = False      # (split rows: workspace matrices change layout with the mode)
    check(lib().cra5_hyper_attention_f32(_p(qkv), _p(out), _p(out_split.data) if out_split is not None else None,
                                         out_split.Kp if out_split is not None else 0, n, C, heads,
                                         float((C // heads) ** -0.5), _stream()), "cra5_hyper_attention_f32")
    if ev is not None:
        TIMER.stop("hyper_attention_f32", ev, 4.0 * n * n * C)
    return out if out is not None else out_split


def layernorm(x, gamma, beta, eps=1e-6, out=None, out_split=None, want_f32=True, out_plain=False):
    _dev(x, gamma, beta, out)
    rows, D = x.shape
    if out is None and want_f32:
        out = torch.empty((rows, D), device=x.device, dtype=torch.float32)
    if out_split is not None:
        assert out_split.rows == rows and out_split.K == D
    ev = TIMER.start() if TIMER is not None else None
    check(lib().cra5_layernorm_f32(_p(x), _row_stride(x), _p(gamma), _p(beta), _p(out),
                                   _row_stride(out) if out is not None else 0,
                                   _p(out_split.data) if out_split is not None else None,
                                   out_split.Kp if out_split is not None else 0, rows, D, float(eps),
                                   int(bool(out_plain)), _stream()),
          "cra5_layernorm_f32")
    if out_split is not None:
        out_split.plain = bool(out_plain)
    if ev is not None:   # bytes: the row read once + every output written once
        # (the 648-row hyper-prior LayerNorms are launch-latency-bound: kept out of the HBM-bound figure)
        TIMER.stop("layernorm" if rows >= 4096 else "layernorm_small", ev,
                   4.0 * rows * D * (1 + (out is not None) + (out_split is not None)))
    return out if out is not None else out_split


def window_attention(qkv, pad_row, heads, H, W, wh, ww, out=None, out_split=None, want_f32=True):
    """qkv: [H*W, 3C] contiguous; returns [H*W, C] (fp32) and/or fills out_split (SplitMat,
    whose pad columns must already be zero)."""
    _dev(qkv, pad_row, out)
    N, C3 = qkv.shape
    C = C3 // 3
    assert N == H * W and qkv.is_contiguous() and pad_row.numel() == C3
    if out is None and want_f32:
        out = torch.empty((N, C), device=qkv.device, dtype=torch.float32)
    assert out is None or out.is_contiguous()
    if out_split is not None:
        assert out_split.rows == N and out_split.K == C
    scale = float((C // heads) ** -0.5)
    ev = TIMER.start() if TIMER is not None else None
    if out_split is not None:
        out_split.plain = False      # (split rows: workspace matrices change layout with the mode)
    check(lib().cra5_window_attention_f32(_p(qkv), _p(pad_row), _p(out),
                                          _p(out_split.data) if out_split is not None else None,
                                          out_split.Kp if out_split is not None else 0, C, heads, H, W, wh, ww,
                                          scale, _stream()), "cra5_window_attention_f32")
    if ev is not None:
        # algorithmic flops: real (unpadded) queries x all keys of their window, QK^T + PV
        L = wh * ww
        TIMER.stop("window_attention_f32", ev, 4.0 * N * L * C)
    return out if out is not None else out_split


def attention_workspace_bytes(n_tokens, heads):
    """Bytes of device workspace the balanced whole-grid attention schedule wants for this shape (0: none)."""
    return int(lib().cra5_attention_workspace_bytes(int(n_tokens), int(heads)))


def attention_balanced_plan(n_tokens, heads):
    """(plan exists, workspace bytes it needs - possibly 0) of the balanced whole-grid attention schedule on this device."""
    nb = ctypes.c_size_t(0)
    ok = lib().cra5_attention_balanced_plan(int(n_tokens), int(heads), ctypes.byref(nb))
    return bool(ok), int(nb.value)


def window_attention_split(qkv_s, pad_s, heads, H, W, wh, ww, out=None, out_split=None, hi_only=False, workspace=None,
                           balanced=None, persistent_units=False):
    """qkv_s: SplitMat [H*W, 3C]; pad_s: SplitMat [1, 3C] (the split qkv bias).  workspace: a device byte tensor
    of >= attention_workspace_bytes(H*W, heads) -> the balanced schedule for whole-grid launches (balanced=True with
    workspace=None: a plan that needs no workspace)."""
    _devs(qkv_s.data, pad_s.data)
    _dev(out)
    N, C = qkv_s.rows, qkv_s.K // 3
    assert N == H * W and qkv_s.Kp == 3 * C and pad_s.Kp == 3 * C
    if out_split is not None:
        assert out_split.rows == N and out_split.K == C
    scale = float((C // heads) ** -0.5)
    # reduced-precision mode on PLAIN rows (hi_only = 3 in the C ABI): the qkv GEMM wrote them plain; the pad row must be
    # plain too and out_split is written plain
    hi_flag = int(bool(hi_only))
    if qkv_s.plain:
        assert hi_only and pad_s.plain, "plain qkv rows need the reduced-precision mode and a plain pad row"
        hi_flag = 3
    else:
        assert not pad_s.plain
    if out_split is not None:
        out_split.plain = hi_flag == 3
    if persistent_units:
        hi_flag |= 4          # CRA5_ATTN_PERSISTENT_UNITS: the alternative windowed schedule (measurements, tests)
    ev = TIMER.start() if TIMER is not None else None
    if workspace is not None or balanced:
        assert workspace is None or (workspace.is_cuda and workspace.is_contiguous())
        check(lib().cra5_window_attention_split_ws(_p(qkv_s.data), qkv_s.Kp, _p(pad_s.data), _p(out),
                                                   _p(out_split.data) if out_split is not None else None,
                                                   out_split.Kp if out_split is not None else 0, C, heads, H, W, wh, ww,
                                                   scale, hi_flag,
                                                   ctypes.c_void_p(workspace.data_ptr()) if workspace is not None else None,
                                                   workspace.numel() * workspace.element_size() if workspace is not None else 0,
                                                   _stream()),
              "cra5_window_attention_split_ws")
    else:
        check(lib().cra5_window_attention_split(_p(qkv_s.data), qkv_s.Kp, _p(pad_s.data), _p(out),
                                                _p(out_split.data) if out_split is not None else None,
                                                out_split.Kp if out_split is not None else 0, C, heads, H, W, wh, ww,
                                                scale, hi_flag, _stream()), "cra5_window_attention_split")
    if ev is not None:
        TIMER.stop("window_attention_split", ev, 4.0 * N * (wh * ww) * C)
    return out if out is not None else out_split


MAX_WIN_TOKENS = 1152   # csrc/attention_split_f16.hip: windowed launches keep a per-block row-offset table


def split_attention_ok(C, heads, wh, ww, H=None, W=None):
    """Shapes the f16-MFMA attention kernel takes: head dim 64, window length a multiple of 32,
    and either the whole (H, W) grid as one window or a window of <= MAX_WIN_TOKENS tokens."""
    L = wh * ww
    return (C % heads == 0 and C // heads == 64 and L % 32 == 0
            and (L <= MAX_WIN_TOKENS or (wh == H and ww == W)))


def im2col(x, kh, kw, sh, sw, ldk=None, mean=None, std=None, out=None, out_split=None, out_plain=False):
    """x: [C,H,W] -> cols [Hp*Wp, ldk] (column (c*kh+i)*kw+j); pad columns are zero.  With
    out_split (a ZERO-initialised SplitMat [Hp*Wp, C*kh*kw]) the split-f16 form is written
    instead of the fp32 one."""
    _dev(x, mean, std, out)
    C, H, W = x.shape
    Hp, Wp = (H - kh) // sh + 1, (W - kw) // sw + 1
    K = C * kh * kw
    assert x.is_contiguous()
    if out_split is not None:
        assert out_split.rows == Hp * Wp and out_split.K == K
        ev = TIMER.start() if TIMER is not None else None
        out_split.plain = bool(out_plain)
        check(lib().cra5_im2col_f32(_p(x), _p(mean), _p(std), None, _p(out_split.data), C, H, W, kh, kw, sh, sw, Hp,
                                    Wp, out_split.Kp, int(bool(out_plain)), _stream()), "cra5_im2col_f32")
        if ev is not None:   # bytes: the frame read once + the patch matrix written once
            TIMER.stop("im2col", ev, 4.0 * C * H * W + 4.0 * Hp * Wp * K)
        return out_split
    ldk = ldk or K
    if out is None:
        out = torch.zeros((Hp * Wp, ldk), device=x.device, dtype=torch.float32)
    assert out.is_contiguous() and out.shape[1] == ldk
    check(lib().cra5_im2col_f32(_p(x), _p(mean), _p(std), _p(out), None, C, H, W, kh, kw, sh, sw, Hp, Wp, ldk,
                                0, _stream()), "cra5_im2col_f32")
    return out


def col2im(cols, C, kh, kw, sh, sw, Hp, Wp, mean=None, std=None, out=None):
    _dev(cols, mean, std, out)
    H, W = (Hp - 1) * sh + kh, (Wp - 1) * sw + kw
    if out is None:
        out = torch.empty((C, H, W), device=cols.device, dtype=torch.float32)
    assert out.is_contiguous()
    ev = TIMER.start() if TIMER is not None else None
    check(lib().cra5_col2im_f32(_p(cols), _p(mean), _p(std), _p(out), C, H, W, kh, kw, sh, sw, Hp, Wp,
                                _row_stride(cols), _stream()), "cra5_col2im_f32")
    if ev is not None:   # bytes: the column matrix read once + the frame written once
        TIMER.stop("col2im", ev, 4.0 * Hp * Wp * C * kh * kw + 4.0 * C * H * W)
    return out


PROBE_PARTIALS = 256     # CRA5_PROBE_PARTIALS


def probe_sums(x, out, stride=1):
    """cra5_probe_sums_f32: PROBE_PARTIALS partial sums of the flat fp32 tensor `x` (every `stride`-th element) into the
    fp32 device slice `out` [PROBE_PARTIALS]; non-finite anywhere in the sampled elements <=> a non-finite partial."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and out.is_cuda and out.dtype == torch.float32
            and out.is_contiguous() and out.numel() == PROBE_PARTIALS):
        raise TypeError("probe_sums takes a contiguous fp32 device tensor and a contiguous fp32 device slice of PROBE_PARTIALS")
    check(lib().cra5_probe_sums_f32(_p(x), x.numel(), int(stride), _p(out), _stream()), "cra5_probe_sums_f32")
    return out


def transpose(x, out=None):
    _dev(x, out)
    R, Cc = x.shape
    if out is None:
        out = torch.empty((Cc, R), device=x.device, dtype=torch.float32)
    check(lib().cra5_transpose_f32(_p(x), _row_stride(x), _p(out), _row_stride(out), R, Cc, _stream()),
          "cra5_transpose_f32")
    return out


def pixel_shuffle(lin, Hz, Wz, p1, p2, out=None):
    _dev(lin, out)
    F = lin.shape[1]
    Cout = F // (p1 * p2)
    assert lin.is_contiguous() and lin.shape[0] == Hz * Wz
    if out is None:
        out = torch.empty((Cout, Hz * p1, Wz * p2), device=lin.device, dtype=torch.float32)
    check(lib().cra5_pixel_shuffle_f32(_p(lin), _p(out), Hz, Wz, p1, p2, Cout, _stream()), "cra5_pixel_shuffle_f32")
    return out


def conv2d(x, w_split, bias, k, stride, act=None):
    """nn.Conv2d(k, stride, padding=k//2) on one image x [C, H, W] (models/utils.py:128-135): padded patch
    gather -> split-f16 GEMM (+ bias) -> [Cout, Ho, Wo].  w_split: SplitMat of weight.reshape(Cout, -1)."""
    _dev(x, bias)
    C, H, W = x.shape
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    cols = SplitMat.empty(Ho * Wo, C * k * k, x.device)
    check(lib().cra5_conv_im2col_f32(_p(x.contiguous()), _p(cols.data), C, H, W, k, k, stride, stride, p, p, Ho, Wo,
                                     cols.Kp, _stream()), "cra5_conv_im2col_f32")
    tok = gemm_nt_split(cols, w_split, bias=bias)                 # [Ho*Wo, Cout]
    out = transpose(tok).view(w_split.rows, Ho, Wo)
    return unary(out, act) if act else out


def conv_transpose2d(x, w_split, bias, cout, k, stride):
    """nn.ConvTranspose2d(k, stride, padding=k//2, output_padding=stride-1) on x [Cin, Hi, Wi]
    (models/utils.py:138-146).  w_split: SplitMat of weight.reshape(Cin, cout*k*k).t()."""
    _dev(x, bias)
    Cin, Hi, Wi = x.shape
    p = k // 2
    Ho, Wo = (Hi - 1) * stride - 2 * p + k + (stride - 1), (Wi - 1) * stride - 2 * p + k + (stride - 1)
    tok = split_f16(transpose(x.reshape(Cin, Hi * Wi).contiguous()))          # [Hi*Wi, Cin]
    cols = gemm_nt_split(tok, w_split)                                        # [Hi*Wi, cout*k*k]
    out = torch.empty((cout, Ho, Wo), device=x.device, dtype=torch.float32)
    check(lib().cra5_deconv_col2im_f32(_p(cols), _p(bias), _p(out), cout, Hi, Wi, k, k, stride, stride, p, p, Ho, Wo,
                                       _row_stride(cols), _stream()), "cra5_deconv_col2im_f32")
    return out


def unary(x, op, slope=0.0):
    """op: 'relu' | 'leaky_relu' (slope) | 'abs'."""
    _dev(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    code, sl = {"relu": (0, 0.0), "leaky_relu": (0, slope or 0.01), "abs": (1, 0.0)}[op]
    check(lib().cra5_unary_f32(_p(x), _p(y), x.numel(), code, float(sl), _stream()), "cra5_unary_f32")
    return y


def gaussian_conditional(scales, means, scale_table, y=None, sym_in=None, want=("idx", "sym", "y_hat"),
                         scale_bound=0.11, lik_bound=1e-9):
    """Fused GC kernel. Returns dict with the requested outputs (flat views shaped like `means`)."""
    _dev(scales, means, scale_table, y, sym_in)
    assert scales.is_contiguous() and means.is_contiguous()
    n = means.numel()
    dev = means.device
    o = {}
    o["idx"] = torch.empty(means.shape, device=dev, dtype=torch.int32) if "idx" in want else None
    o["sym"] = torch.empty(means.shape, device=dev, dtype=torch.int32) if "sym" in want else None
    o["y_hat"] = torch.empty(means.shape, device=dev, dtype=torch.float32) if "y_hat" in want else None
    o["lik"] = torch.empty(means.shape, device=dev, dtype=torch.float32) if "lik" in want else None
    if y is not None:
        assert y.is_contiguous() and y.numel() == n
    if sym_in is not None:
        assert sym_in.is_contiguous() and sym_in.numel() == n and sym_in.dtype == torch.int32
    n_table = 0 if scale_table is None else scale_table.numel()   # no table: no indexes (forward() before update())
    if n_table == 0 and "idx" in want:
        raise ValueError("gaussian_conditional: indexes requested without a scale table (run update())")
    check(lib().cra5_gaussian_conditional_f32(_p(y), _p(sym_in), _p(scales), _p(means), _p(scale_table),
                                              n_table, float(scale_bound), float(lik_bound), _p(o["idx"]),
                                              _p(o["sym"]), _p(o["y_hat"]), _p(o["lik"]), n, _stream()),
          "cra5_gaussian_conditional_f32")
    return {k: v for k, v in o.items() if v is not None}


def entropy_bottleneck(medians, params, z=None, sym_in=None, want=("sym", "z_hat"), lik_bound=1e-9, shape=None, sym_out=None):
    """z / sym_in: [C, n] (channel-major).  sym_out: caller-owned int32 device tensor for the symbols (a slice of a packed
    device -> host record buffer)."""
    _dev(medians, params, z, sym_in)
    src = z if z is not None else sym_in
    C = medians.numel()
    n_per = src.numel() // C
    dev = src.device
    o = {}
    o["sym"] = (sym_out.view(src.shape) if sym_out is not None else torch.empty(src.shape, device=dev, dtype=torch.int32)) if "sym" in want else None
    if sym_out is not None:
        assert sym_out.dtype == torch.int32 and sym_out.is_contiguous() and sym_out.numel() == src.numel()
    o["z_hat"] = torch.empty(src.shape, device=dev, dtype=torch.float32) if "z_hat" in want else None
    o["lik"] = torch.empty(src.shape, device=dev, dtype=torch.float32) if "lik" in want else None
    assert src.is_contiguous()
    check(lib().cra5_entropy_bottleneck_f32(_p(z), _p(sym_in), _p(medians), _p(params), float(lik_bound), _p(o["sym"]),
                                            _p(o["z_hat"]), _p(o["lik"]), C, n_per, _stream()),
          "cra5_entropy_bottleneck_f32")
    return {k: v for k, v in o.items() if v is not None}


def gdn(x, beta_eff, gamma_eff, inverse=False):
    _dev(x, beta_eff, gamma_eff)
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    check(lib().cra5_gdn_f32(_p(x), _p(beta_eff), _p(gamma_eff), _p(y), B, C, HW, int(bool(inverse)), _stream()),
          "cra5_gdn_f32")
    return y


# ------------------------------------------------------------------ host entropy coding


def _np_i32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int32)


def rans_encode(symbols, indexes, cdf, cdf_len, offsets):
    """-> bytes. Arguments: int32 arrays (numpy or CPU tensors); cdf is [n_cdfs, stride]."""
    s, i = _np_i32(symbols).reshape(-1), _np_i32(indexes).reshape(-1)
    c, l, o = _np_i32(cdf), _np_i32(cdf_len).reshape(-1), _np_i32(offsets).reshape(-1)
    if s.size != i.size:
        raise ValueError("`symbols` and `indexes` should have the same size.")
    out = ctypes.c_void_p()
    n = ctypes.c_size_t()
    check(lib().cra5_rans_encode_with_indexes(s.ctypes.data, i.ctypes.data, s.size, c.ctypes.data, c.shape[0],
                                              c.shape[1], l.ctypes.data, o.ctypes.data, ctypes.byref(out),
                                              ctypes.byref(n)), "cra5_rans_encode_with_indexes")
    try:
        return ctypes.string_at(out.value, n.value)
    finally:
        lib().cra5_free(out)


def rans_resolve_symbols(symbols, indexes, cdf, cdf_len, offsets, out=None):
    """Device side of the resolved encoder: int32 DEVICE tensors symbols / indexes [n], tables
    cdf [n_cdfs, stride], cdf_len, offsets -> (start_range uint32-as-int32 [n], raw [n], esc uint8 [n])."""
    for t in (symbols, indexes, cdf, cdf_len, offsets):
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise TypeError("rans_resolve_symbols takes contiguous int32 device tensors")
    n = symbols.numel()
    if indexes.numel() != n:
        raise ValueError("`symbols` and `indexes` should have the same size.")
    if out is None:
        out = (torch.empty(n, device=symbols.device, dtype=torch.int32),
               torch.empty(n, device=symbols.device, dtype=torch.int32),
               torch.empty(n, device=symbols.device, dtype=torch.uint8))
    sr, raw, esc = out
    check(lib().cra5_rans_resolve_symbols_i32(_p(symbols), _p(indexes), n, _p(cdf), cdf.shape[0], cdf.shape[1],
                                              _p(cdf_len), _p(offsets), _p(sr), _p(raw),
                                              ctypes.c_void_p(esc.data_ptr()), _stream()),
          "cra5_rans_resolve_symbols_i32")
    return sr, raw, esc


def rans_encode_resolved(start_range, raw, esc):
    """-> bytes, from HOST arrays (numpy / CPU tensors): start_range, raw 32-bit words, esc uint8."""
    def _np(a, dt):
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(a).view(dt) if a.dtype.itemsize == np.dtype(dt).itemsize else np.ascontiguousarray(a, dtype=dt)
    s, r, e = _np(start_range, np.uint32).reshape(-1), _np(raw, np.uint32).reshape(-1), _np(esc, np.uint8).reshape(-1)
    if not (s.size == r.size == e.size):
        raise ValueError("start_range, raw and esc must have the same size")
    out = ctypes.c_void_p()
    n = ctypes.c_size_t()
    check(lib().cra5_rans_encode_resolved(s.ctypes.data, r.ctypes.data, e.ctypes.data, s.size, ctypes.byref(out),
                                          ctypes.byref(n)), "cra5_rans_encode_resolved")
    try:
        return ctypes.string_at(out.value, n.value)
    finally:
        lib().cra5_free(out)


def rans_decode(data, indexes, cdf, cdf_len, offsets, out=None):
    """-> int32 numpy array of len(indexes) (written into `out`, a contiguous int32 array of that
    size, when given: e.g. the numpy view of a pinned staging tensor)."""
    i = _np_i32(indexes).reshape(-1)
    c, l, o = _np_i32(cdf), _np_i32(cdf_len).reshape(-1), _np_i32(offsets).reshape(-1)
    if out is None:
        out = np.empty(i.size, dtype=np.int32)
    elif not (isinstance(out, np.ndarray) and out.dtype == np.int32 and out.flags.c_contiguous and out.size == i.size):
        raise ValueError("`out` must be a contiguous int32 numpy array with one entry per index")
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    check(lib().cra5_rans_decode_with_indexes(ctypes.addressof(buf), len(data), i.ctypes.data, i.size, c.ctypes.data,
                                              c.shape[0], c.shape[1], l.ctypes.data, o.ctypes.data, out.ctypes.data),
          "cra5_rans_decode_with_indexes")
    return out


def rans_resolve_symbols_compact(symbols, indexes, cdf, cdf_len, offsets, out=None):
    """Device side of the compact resolved encoder -> (start_range int32 [n], rec16 int16 [n], overflow int32 [1]).
    out: caller-owned (sr, rec, ovf) device tensors of those types (slices of a packed record buffer)."""
    for t in (symbols, indexes, cdf, cdf_len, offsets):
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise TypeError("rans_resolve_symbols_compact takes contiguous int32 device tensors")
    n = symbols.numel()
    if indexes.numel() != n:
        raise ValueError("`symbols` and `indexes` should have the same size.")
    if out is not None:
        sr, rec, ovf = out
        assert sr.dtype == torch.int32 and rec.dtype == torch.int16 and ovf.dtype == torch.int32
        assert sr.numel() == n and rec.numel() == n and ovf.numel() == 1 and sr.is_contiguous() and rec.is_contiguous()
    else:
        sr = torch.empty(n, device=symbols.device, dtype=torch.int32)
        rec = torch.empty(n, device=symbols.device, dtype=torch.int16)
        ovf = torch.empty(1, device=symbols.device, dtype=torch.int32)
    check(lib().cra5_rans_resolve_symbols_compact(_p(symbols), _p(indexes), n, _p(cdf), cdf.shape[0], cdf.shape[1],
                                                  _p(cdf_len), _p(offsets), _p(sr), ctypes.c_void_p(rec.data_ptr()),
                                                  _p(ovf), _stream()), "cra5_rans_resolve_symbols_compact")
    return sr, rec, ovf


def rans_encode_resolved_compact(start_range, rec16):
    """-> bytes, from HOST arrays: start_range (32-bit words), rec16 (16-bit records)."""
    s_ = np.ascontiguousarray(start_range).view(np.uint32).reshape(-1)
    r_ = np.ascontiguousarray(rec16).view(np.uint16).reshape(-1)
    if s_.size != r_.size:
        raise ValueError("start_range and rec16 must have the same size")
    out = ctypes.c_void_p()
    n = ctypes.c_size_t()
    check(lib().cra5_rans_encode_resolved_compact(s_.ctypes.data, r_.ctypes.data, s_.size, ctypes.byref(out),
                                                  ctypes.byref(n)), "cra5_rans_encode_resolved_compact")
    try:
        return ctypes.string_at(out.value, n.value)
    finally:
        lib().cra5_free(out)


def rans_decode_compact(data, indexes_u8, cdf, cdf_len, offsets, out):
    """cra5_rans_decode_with_indexes_u8_i16: uint8 index array -> the int16 numpy array `out` (e.g. the view of a pinned
    staging tensor).  Raises Cra5Error with status ERR_RANGE when a symbol does not fit int16 (decode again with
    rans_decode)."""
    if not (isinstance(indexes_u8, np.ndarray) and indexes_u8.dtype == np.uint8 and indexes_u8.flags.c_contiguous):
        raise ValueError("`indexes_u8` must be a contiguous uint8 numpy array")
    if not (isinstance(out, np.ndarray) and out.dtype == np.int16 and out.flags.c_contiguous and out.size == indexes_u8.size):
        raise ValueError("`out` must be a contiguous int16 numpy array with one entry per index")
    c, l, o = _np_i32(cdf), _np_i32(cdf_len).reshape(-1), _np_i32(offsets).reshape(-1)
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    check(lib().cra5_rans_decode_with_indexes_u8_i16(ctypes.addressof(buf), len(data), indexes_u8.ctypes.data,
                                                     indexes_u8.size, c.ctypes.data, c.shape[0], c.shape[1],
                                                     l.ctypes.data, o.ctypes.data, out.ctypes.data),
          "cra5_rans_decode_with_indexes_u8_i16")
    return out


def gaussian_conditional_compact(scales, means, scale_table=None, sym16_in=None, want_idx8=False, scale_bound=0.11, idx8_out=None):
    """Decode-side halves of gaussian_conditional on compact records: want_idx8 -> uint8 CDF indexes (same shape as
    means); sym16_in (int16 device tensor) -> y_hat = sym + mean (fp32).  Returns {"idx8": ..., "y_hat": ...}."""
    _dev(scales, means, scale_table, sym16_in)
    n = means.numel()
    out = {}
    idx8 = (idx8_out if idx8_out is not None else torch.empty(means.shape, device=means.device, dtype=torch.uint8)) if want_idx8 else None
    if idx8_out is not None:
        assert idx8_out.dtype == torch.uint8 and idx8_out.is_contiguous() and idx8_out.numel() == n
    y_hat = torch.empty(means.shape, device=means.device, dtype=torch.float32) if sym16_in is not None else None
    if sym16_in is not None:
        assert sym16_in.dtype == torch.int16 and sym16_in.is_contiguous() and sym16_in.numel() == n
    assert means.is_contiguous() and (scales is None or scales.is_contiguous())
    check(lib().cra5_gaussian_conditional_compact_f32(
        _p(scales), _p(means), _p(scale_table), scale_table.numel() if scale_table is not None else 0, float(scale_bound),
        ctypes.c_void_p(sym16_in.data_ptr()) if sym16_in is not None else None,
        ctypes.c_void_p(idx8.data_ptr()) if idx8 is not None else None, _p(y_hat), n, _stream()),
          "cra5_gaussian_conditional_compact_f32")
    if idx8 is not None:
        out["idx8"] = idx8
    if y_hat is not None:
        out["y_hat"] = y_hat
    return out


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.empty(p.size + 1, dtype=np.uint32)
    rc = lib().cra5_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data)
    if rc in (-3, -4, -5):
        raise ValueError({-3: "Invalid `pmf`, non-finite or negative element found",
                          -4: "Invalid `pmf`: at least one element must have a non-zero probability.",
                          -5: "Invalid `pmf`: no bin can donate frequency"}[rc])
    check(rc, "cra5_pmf_to_quantized_cdf")
    return out
