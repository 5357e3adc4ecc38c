"""Drop-in for the reference's pybind11 module `compressai.ans`
(cra5/models/compressai/cpp_exts/rans/rans_interface.cpp:361-381): same three classes, same
method names and argument meaning (Python lists / bytes in and out), backed by the C ABI of
libcra5_amd.so.  `import cra5_amd.ans as ans; sys.modules["compressai.ans"] = ans` is all a
reference checkout needs."""
import ctypes

import numpy as np

from . import ops
from ._lib import check, lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _tables(cdfs, cdfs_sizes, offsets):
    sizes = _i32(cdfs_sizes).reshape(-1)
    if isinstance(cdfs, np.ndarray) and cdfs.ndim == 2:
        c = _i32(cdfs)
    else:  # list of (possibly ragged) rows, as pybind11 accepts
        stride = max(len(r) for r in cdfs)
        c = np.zeros((len(cdfs), stride), dtype=np.int32)
        for i, r in enumerate(cdfs):
            c[i, : len(r)] = r
    return c, sizes, _i32(offsets).reshape(-1)


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        c, l, o = _tables(cdfs, cdfs_sizes, offsets)
        return ops.rans_encode(_i32(symbols), _i32(indexes), c, l, o)


class BufferedRansEncoder:
    def __init__(self):
        self._h = ctypes.c_void_p(lib().cra5_rans_encoder_new())
        if not self._h:
            raise MemoryError("cra5_rans_encoder_new")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cra5_rans_encoder_free(self._h)
            self._h = None

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        s, i = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        if s.size != i.size:
            raise ValueError("`symbols` and `indexes` should have the same size.")
        c, l, o = _tables(cdfs, cdfs_sizes, offsets)
        check(lib().cra5_rans_encoder_push(self._h, s.ctypes.data, i.ctypes.data, s.size, c.ctypes.data, c.shape[0],
                                           c.shape[1], l.ctypes.data, o.ctypes.data), "cra5_rans_encoder_push")

    def flush(self):
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().cra5_rans_encoder_flush(self._h, ctypes.byref(out), ctypes.byref(n)), "cra5_rans_encoder_flush")
        try:
            return ctypes.string_at(out.value, n.value)
        finally:
            lib().cra5_free(out)


class RansDecoder:
    def __init__(self):
        self._h = ctypes.c_void_p(lib().cra5_rans_decoder_new())
        if not self._h:
            raise MemoryError("cra5_rans_decoder_new")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cra5_rans_decoder_free(self._h)
            self._h = None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        c, l, o = _tables(cdfs, cdfs_sizes, offsets)
        return ops.rans_decode(encoded, _i32(indexes), c, l, o).tolist()

    def set_stream(self, encoded):
        buf = (ctypes.c_char * len(encoded)).from_buffer_copy(encoded)
        check(lib().cra5_rans_decoder_set_stream(self._h, ctypes.addressof(buf), len(encoded)),
              "cra5_rans_decoder_set_stream")

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        i = _i32(indexes).reshape(-1)
        c, l, o = _tables(cdfs, cdfs_sizes, offsets)
        out = np.empty(i.size, dtype=np.int32)
        check(lib().cra5_rans_decoder_decode_stream(self._h, i.ctypes.data, i.size, c.ctypes.data, c.shape[0],
                                                    c.shape[1], l.ctypes.data, o.ctypes.data, out.ctypes.data),
              "cra5_rans_decoder_decode_stream")
        return out.tolist()


def pmf_to_quantized_cdf(pmf, precision):
    """`compressai._CXX.pmf_to_quantized_cdf` (ops.cpp:112-118): list in, list out."""
    return ops.pmf_to_quantized_cdf(np.asarray(pmf, dtype=np.float32), precision).tolist()
