"""TEST INFRASTRUCTURE ONLY (oracle) - never imported by the product path.

Pure-Python restatement of the reference's rANS64 coder, exposing the same three
class names as the pybind11 module ``compressai.ans``
(/root/reference/cra5/models/compressai/cpp_exts/rans/rans_interface.cpp:361-381).

Parity status: **parity unpinned** at the byte level.  The reference's module
``#include "rans64.h"`` (rans_interface.hpp:36) from ``third_party/ryg_rans`` - an
un-vendored submodule of upstream CompressAI (ryg_rans by F. Giesen, public domain,
no version pin in the reference tree) - so the reference extension cannot be built
here and the reference holds no known-answer bitstream.  What IS pinned by source:
  * the wrapper logic (escape / bypass, reverse-order flush, 32-bit word stream):
    rans_interface.cpp:69-105 (bypass put/get), :108-173 (symbol buffering),
    :175-200 (flush), :215-284 (decode);
  * the six rans64.h primitives, restated from their published public-domain
    definition (RANS64_L = 1<<31, 32-bit renormalisation) in the docstrings below.
Integers are Python ints masked to 64 bits, so this file is an implementation
independent of oracle/rans_ref.c (C) and of the product's C++ coder.
"""
import struct

PRECISION = 16          # rans_interface.cpp:49
BYPASS_PRECISION = 4    # rans_interface.cpp:51
MAX_BYPASS_VAL = (1 << BYPASS_PRECISION) - 1   # rans_interface.cpp:52
RANS64_L = 1 << 31      # rans64.h: lower bound of the normalisation interval
M64 = (1 << 64) - 1


def buffer_symbols(symbols, indexes, cdfs, cdfs_sizes, offsets, out):
    """rans_interface.cpp:108-173. Appends (start, range, bypass) triples."""
    for i in range(len(symbols)):
        cdf_idx = indexes[i]
        cdf = cdfs[cdf_idx]
        max_value = cdfs_sizes[cdf_idx] - 2
        value = symbols[i] - offsets[cdf_idx]
        raw_val = 0
        if value < 0:
            raw_val = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw_val = 2 * (value - max_value)
            value = max_value
        out.append((cdf[value] & 0xFFFF, (cdf[value + 1] - cdf[value]) & 0xFFFF, False))
        if value == max_value:
            n_bypass = 0
            while (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS_VAL:
                out.append((MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, True))
                val -= MAX_BYPASS_VAL
            out.append((val, val + 1, True))
            for j in range(n_bypass):
                v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL
                out.append((v, v + 1, True))


def flush_symbols(syms):
    """rans_interface.cpp:175-200 with the rans64.h primitives:

    Rans64EncInit:  x = RANS64_L
    Rans64EncPut(start, freq, bits):
        x_max = ((RANS64_L >> bits) << 32) * freq
        if x >= x_max: emit low 32 bits (written *downwards*), x >>= 32
        x = ((x // freq) << bits) + (x % freq) + start
    Rans64EncPutBits(val, nbits) (rans_interface.cpp:69-87):
        freq = 1 << (16 - nbits); x_max = ((RANS64_L >> 16) << 32) * freq
        same renormalisation; x = (x << nbits) | val
    Rans64EncFlush: two words, low word first in memory.
    Output = the words from the final pointer to the end of the buffer, i.e. in
    REVERSE order of emission, host little-endian.
    """
    x = RANS64_L
    words = []  # emission order; memory order is the reverse
    for start, rng, bypass in reversed(syms):
        if not bypass:
            x_max = ((RANS64_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (((x // rng) << PRECISION) + (x % rng) + start) & M64
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS64_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x << BYPASS_PRECISION) | start) & M64
    # flush: ptr -= 2; ptr[0] = low; ptr[1] = high
    words.append((x >> 32) & 0xFFFFFFFF)
    words.append(x & 0xFFFFFFFF)
    words.reverse()
    return struct.pack("<%dI" % len(words), *words)


class _WordReader:
    def __init__(self, data):
        n = len(data) // 4
        self.w = struct.unpack("<%dI" % n, data[: 4 * n])
        self.p = 0

    def next(self):
        v = self.w[self.p]
        self.p += 1
        return v


def _dec_init(rd):
    """Rans64DecInit: x = ptr[0] | ptr[1] << 32; ptr += 2."""
    lo = rd.next()
    hi = rd.next()
    return lo | (hi << 32)


def _dec_get_bits(x, rd, nbits):
    """rans_interface.cpp:89-105."""
    val = x & ((1 << nbits) - 1)
    x >>= nbits
    if x < RANS64_L:
        x = (x << 32) | rd.next()
    return x, val


def decode_symbols(x, rd, indexes, cdfs, cdfs_sizes, offsets):
    """rans_interface.cpp:231-281 (== :306-356).

    Rans64DecGet(bits)      = x & ((1 << bits) - 1)
    Rans64DecAdvance(start, freq, bits):
        x = freq * (x >> bits) + (x & mask) - start
        if x < RANS64_L: x = (x << 32) | *ptr++
    """
    out = []
    mask = (1 << PRECISION) - 1
    for cdf_idx in indexes:
        cdf = cdfs[cdf_idx]
        n = cdfs_sizes[cdf_idx]
        max_value = n - 2
        offset = offsets[cdf_idx]
        cum_freq = x & mask
        # std::find_if(cdf.begin(), cdf.begin()+n, v > cum_freq) - linear scan
        s = 0
        while s < n and not (cdf[s] > cum_freq):
            s += 1
        s -= 1
        start = cdf[s]
        freq = cdf[s + 1] - cdf[s]
        x = freq * (x >> PRECISION) + (x & mask) - start
        if x < RANS64_L:
            x = (x << 32) | rd.next()
        value = s
        if value == max_value:
            x, val = _dec_get_bits(x, rd, BYPASS_PRECISION)
            n_bypass = val
            while val == MAX_BYPASS_VAL:
                x, val = _dec_get_bits(x, rd, BYPASS_PRECISION)
                n_bypass += val
            raw_val = 0
            for j in range(n_bypass):
                x, val = _dec_get_bits(x, rd, BYPASS_PRECISION)
                raw_val |= val << (j * BYPASS_PRECISION)
            value = raw_val >> 1
            if raw_val & 1:
                value = -value - 1
            else:
                value += max_value
        out.append(value + offset)
    return x, out


class BufferedRansEncoder:
    def __init__(self):
        self._syms = []

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        buffer_symbols(symbols, indexes, cdfs, cdfs_sizes, offsets, self._syms)

    def flush(self):
        out = flush_symbols(self._syms)
        self._syms = []
        return out


class RansEncoder:
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        e = BufferedRansEncoder()
        e.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return e.flush()


class RansDecoder:
    def __init__(self):
        self._rd = None
        self._x = None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        rd = _WordReader(encoded)
        x = _dec_init(rd)
        _, out = decode_symbols(x, rd, indexes, cdfs, cdfs_sizes, offsets)
        return out

    def set_stream(self, encoded):
        self._rd = _WordReader(encoded)
        self._x = _dec_init(self._rd)

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        self._x, out = decode_symbols(self._x, self._rd, indexes, cdfs, cdfs_sizes, offsets)
        return out
