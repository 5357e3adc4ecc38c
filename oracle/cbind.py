"""TEST INFRASTRUCTURE ONLY (oracle): ctypes binding of oracle/liboracle.so
(rans_ref.c, pmf_ref.c) and of oracle/_ref/_CXX (the reference's ops.cpp)."""
import ctypes
import glob
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.oracle_rans_encode.argtypes = [i32p, i32p, ctypes.c_size_t, i32p, ctypes.c_int, ctypes.c_int, i32p, i32p,
                                         ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t)]
        L.oracle_rans_encode.restype = ctypes.c_int
        L.oracle_rans_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, i32p, ctypes.c_size_t, i32p, ctypes.c_int,
                                         ctypes.c_int, i32p, i32p, i32p]
        L.oracle_rans_decode.restype = ctypes.c_int
        L.oracle_free.argtypes = [ctypes.c_void_p]
        L.oracle_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                                  ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        _lib = L
    return _lib


def _i32(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def rans_encode(symbols, indexes, cdf, cdf_len, off):
    s, i, c, l, o = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1), _i32(cdf), _i32(cdf_len), _i32(off)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    rc = lib().oracle_rans_encode(_p(s), _p(i), s.size, _p(c), c.shape[0], c.shape[1], _p(l), _p(o),
                                  ctypes.byref(out), ctypes.byref(n))
    if rc:
        raise RuntimeError(f"oracle_rans_encode rc={rc}")
    b = ctypes.string_at(out, n.value)
    lib().oracle_free(out)
    return b


def rans_decode(data, indexes, cdf, cdf_len, off):
    import torch
    i, c, l, o = _i32(indexes).reshape(-1), _i32(cdf), _i32(cdf_len), _i32(off)
    out = np.empty(i.size, dtype=np.int32)
    rc = lib().oracle_rans_decode(data, len(data), _p(i), i.size, _p(c), c.shape[0], c.shape[1], _p(l), _p(o), _p(out))
    if rc:
        raise RuntimeError(f"oracle_rans_decode rc={rc}")
    return torch.from_numpy(out)


def pmf_to_cdf(pmf, precision=16):
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.empty(p.size + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(p.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), p.size, precision,
                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    if rc:
        raise ValueError(f"invalid pmf (rc={rc})")
    return out


def ref_cxx():
    """The reference's own ops.cpp, compiled into oracle/_ref (None if absent)."""
    cands = glob.glob(os.path.join(_HERE, "_ref", "_CXX*.so"))
    if not cands:
        return None
    spec = importlib.util.spec_from_file_location("_CXX", cands[0])
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
