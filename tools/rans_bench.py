"""Host rANS coder on one frame's worth of symbols (256 x 72 x 144, default GC tables), the frame path's entry points:
cra5_rans_encode_resolved_compact, cra5_rans_encode_with_indexes, cra5_rans_decode_with_indexes_u8_i16.  CPU only.

  python tools/rans_bench.py LIB.so [LIB2.so ...]     (host-only builds of csrc/host_entropy.cpp, or the product library)

Streams: "default" = build_variants/frame_symbols.npz when present (a real frame of the default synthetic-weight model:
4.3 MB, 37 % of the symbols escape-coded), "matched" = a seeded stream in a trained model's regime (1 MB, no escapes).
Every library must write the same bytes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from cra5_amd.entropy import GaussianConditional, get_scale_table
gc = GaussianConditional(get_scale_table().tolist()); gc.update_scale_table(get_scale_table().tolist(), force=True)
cdf, length, offset = gc.host_tables()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = os.path.join(ROOT, "build_variants", "frame_symbols.npz")
if os.path.exists(f):
    d = np.load(f)
    sym, idx = np.ascontiguousarray(d["y_sym"].astype(np.int32)), np.ascontiguousarray(d["idx"].astype(np.int32))
else:
    rng0 = np.random.default_rng(1)
    idx = np.where(rng0.random(256 * 72 * 144) < 0.5, 0, np.clip(np.rint(rng0.normal(24, 3, size=256 * 72 * 144)), 0, 63)).astype(np.int32)
    sym = np.rint(rng0.standard_normal(idx.size) * np.maximum(np.asarray(get_scale_table())[idx], 30.0)).astype(np.int32)
n = sym.size
def matched():
    rng = np.random.default_rng(0)
    i2 = rng.integers(8, 40, size=n).astype(np.int32)
    s2 = np.rint(rng.standard_normal(n) * np.asarray(get_scale_table())[i2]).astype(np.int32)
    return s2, i2
def resolve_np(sym, idx):
    mx = length[idx] - 2
    v = sym - offset[idx]
    raw = np.where(v < 0, -2 * v - 1, np.where(v >= mx, 2 * (v - mx), 0)).astype(np.int64)
    esc = (v < 0) | (v >= mx)
    vv = np.where(esc, mx, v)
    st = cdf[idx, vv].astype(np.uint32) & 0xFFFF
    rg = (cdf[idx, vv + 1] - cdf[idx, vv]).astype(np.uint32) & 0xFFFF
    sr = (st | (rg << 16)).astype(np.uint32)
    nn = np.zeros(n, np.int64)
    for k in range(1, 9):
        nn = np.where((raw >> (4 * (k - 1))) != 0, k, nn)
    rec = np.where(esc, np.where(raw < 4096, ((nn + 1) << 12) | raw, 0xFFFF), 0).astype(np.uint16)
    return np.ascontiguousarray(sr), np.ascontiguousarray(rec)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
libs = sys.argv[1:] or [os.path.join(ROOT, 'cra5_amd', 'libcra5_amd.so')]
ref = {}
for name, (s_, i_) in (("default", (sym, idx)), ("matched", matched())):
    sr, rec = resolve_np(s_, i_)
    i8 = np.ascontiguousarray(i_.astype(np.uint8))
    for path in libs:
        L = ctypes.CDLL(path)
        out = ctypes.c_void_p(); ln = ctypes.c_size_t()
        be = bd = bf = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            rc = L.cra5_rans_encode_resolved_compact(p(sr), p(rec), ctypes.c_size_t(n), ctypes.byref(out), ctypes.byref(ln))
            be = min(be, time.perf_counter() - t0); assert rc == 0, rc
            enc = ctypes.string_at(out.value, ln.value); L.cra5_free(out)
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.cra5_rans_encode_with_indexes(p(s_), p(i_), ctypes.c_size_t(n), p(cdf), cdf.shape[0], cdf.shape[1], p(length), p(offset), ctypes.byref(out), ctypes.byref(ln))
            bf = min(bf, time.perf_counter() - t0); assert rc == 0
            enc2 = ctypes.string_at(out.value, ln.value); L.cra5_free(out)
        assert enc == enc2
        ref.setdefault(name, enc); assert enc == ref[name], "streams differ between libraries"
        res = np.empty(n, np.int16)
        for _ in range(5):
            t0 = time.perf_counter()
            rc = L.cra5_rans_decode_with_indexes_u8_i16(enc, ctypes.c_size_t(len(enc)), p(i8), ctypes.c_size_t(n), p(cdf), cdf.shape[0], cdf.shape[1], p(length), p(offset), p(res))
            bd = min(bd, time.perf_counter() - t0)
        assert rc == 0 and (res == s_).all(), rc
        print(f"{name:8s} {os.path.basename(path):22s} enc(resolved) {be*1e3:6.1f} ms  enc(full) {bf*1e3:6.1f} ms  decode {bd*1e3:6.1f} ms  ({len(enc)} bytes, escapes {int((rec!=0).sum())})", flush=True)
