"""Host rANS coder micro-benchmark on one frame's worth of symbols (256 x 72 x 144, GC tables).
   python tools/rans_bench.py [lib.so ...]
Symbols: build_variants/frame_symbols.npz when present (a real frame of the synthetic-weight model, dumped by
tools/host_phase_probe.py with CRA5_DUMP_SYMBOLS: 52 % of the symbols sit in the narrowest CDF row and escape),
else a seeded synthetic mix.  Libraries: host-only builds of csrc/host_entropy.cpp (default: the product .so)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the product library is loaded)
from cra5_amd.entropy import GaussianConditional, get_scale_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gc = GaussianConditional(get_scale_table().tolist())
gc.update_scale_table(get_scale_table().tolist(), force=True)
cdf, length, offset = gc.host_tables()
f = os.path.join(ROOT, "build_variants", "frame_symbols.npz")
if os.path.exists(f):
    d = np.load(f)
    sym, idx = np.ascontiguousarray(d["y_sym"].astype(np.int32)), np.ascontiguousarray(d["idx"].astype(np.int32))
else:
    rng = np.random.default_rng(0)
    n = 256 * 72 * 144
    idx = np.where(rng.random(n) < 0.5, 0, np.clip(np.rint(rng.normal(24, 3, size=n)), 0, 63)).astype(np.int32)
    sym = np.rint(rng.standard_normal(n) * np.maximum(np.asarray(get_scale_table())[idx], 1.5)).astype(np.int32)
n = sym.size
libs = sys.argv[1:] or [os.path.join(ROOT, "cra5_amd", "libcra5_amd.so")]
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
ref = None
for rnd in range(2):
    for path in libs:
        L = ctypes.CDLL(path)
        out = ctypes.c_void_p(); ln = ctypes.c_size_t()
        be, bd = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.cra5_rans_encode_with_indexes(p(sym), p(idx), ctypes.c_size_t(n), p(cdf), cdf.shape[0], cdf.shape[1],
                                                 p(length), p(offset), ctypes.byref(out), ctypes.byref(ln))
            be = min(be, time.perf_counter() - t0)
            assert rc == 0
            enc = ctypes.string_at(out.value, ln.value)
            L.cra5_free(out)
        ref = ref or enc
        assert enc == ref, "streams differ between libraries"
        res = np.empty(n, np.int32)
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.cra5_rans_decode_with_indexes(enc, ctypes.c_size_t(len(enc)), p(idx), ctypes.c_size_t(n), p(cdf), cdf.shape[0],
                                                 cdf.shape[1], p(length), p(offset), p(res))
            bd = min(bd, time.perf_counter() - t0)
        assert rc == 0 and (res == sym).all()
        if rnd == 1:
            print(f"{os.path.basename(path):24s} encode {be*1e3:6.1f} ms  decode {bd*1e3:6.1f} ms  ({len(enc)} bytes, {n} symbols)")
