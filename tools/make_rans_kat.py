"""Known-answer generator for the rANS coder (run wherever `pip install compressai` works).

The reference's `compressai.ans` extension cannot be built offline (its `rans64.h` comes from
the un-vendored ryg_rans submodule, rans_interface.hpp:36), so byte-level parity of
cra5_amd's coder with a REAL CompressAI build is "unpinned" until this script has been run
once on a machine that has the upstream wheel:

    pip install compressai            # any version whose ans module has RansEncoder
    python tools/make_rans_kat.py     # writes tests/golden/rans_kat.npz
    git add tests/golden/rans_kat.npz

`tests/test_rans.py::test_known_answer_streams_from_real_compressai` then checks the product
coder (C ABI) and the oracle coders against those streams, both directions.  The inputs are
deterministic (numpy PCG64 seeds below) over the committed table fixture
tests/golden/tables_default.npz; only the `stream_*` arrays depend on compressai.ans.

`--selftest` writes the same file with cra5_amd's own coder instead (marks `source` accordingly)
- useful only to check the plumbing of the test; such a file must NOT be committed.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def cases():
    """-> list of (name, symbols, indexes, cdf [n, stride], cdf_len, offsets), all int32."""
    t = np.load(os.path.join(GOLD, "tables_default.npz"))
    gc = (t["gc_cdf"].astype(np.int32), t["gc_len"].astype(np.int32), t["gc_off"].astype(np.int32))
    eb = (t["eb_cdf"].astype(np.int32), t["eb_len"].astype(np.int32), t["eb_off"].astype(np.int32))
    out = []
    rng = np.random.default_rng(20240601)
    # 1. Gaussian-conditional tables, symbols drawn around zero with the width of their row,
    #    a few percent far outside the table (escape / bypass path, both signs)
    n = 20000
    idx = rng.integers(0, 64, size=n).astype(np.int32)
    width = np.maximum(1.0, (gc[1][idx] - 2) / 8.0)
    sym = np.rint(rng.standard_normal(n) * width).astype(np.int32)
    far = rng.random(n) < 0.03
    sym[far] = (rng.integers(-70000, 70000, size=int(far.sum()))).astype(np.int32)
    out.append(("gc_mixed", sym, idx, *gc))
    # 2. escape-heavy: everything coded in the narrowest row
    n = 5000
    idx = np.zeros(n, dtype=np.int32)
    sym = np.rint(rng.standard_normal(n) * 4).astype(np.int32)
    out.append(("gc_row0_escapes", sym, idx, *gc))
    # 3. factorised (EntropyBottleneck) tables, index = channel id
    C = eb[0].shape[0]
    per = 40
    idx = np.repeat(np.arange(C, dtype=np.int32), per)
    sym = np.rint(rng.standard_normal(C * per) * 3).astype(np.int32)
    out.append(("eb_channels", sym, idx, *eb))
    # 4. degenerate lengths: one symbol, two symbols, one escape with a long payload
    out.append(("one_symbol", np.array([0], np.int32), np.array([5], np.int32), *gc))
    out.append(("two_symbols", np.array([1, -1], np.int32), np.array([63, 0], np.int32), *gc))
    # largest payloads the reference can code: raw < 2^28 (7 nibbles).  Beyond that its nibble-count
    # loop shifts a uint32 by 32 (rans_interface.cpp:152-154: undefined, an endless loop on x86)
    out.append(("big_escape", np.array([2 ** 26, -(2 ** 26)], np.int32), np.array([0, 0], np.int32), *gc))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--selftest", action="store_true", help="use cra5_amd's own coder (plumbing check only)")
    ap.add_argument("--out", default=os.path.join(GOLD, "rans_kat.npz"))
    a = ap.parse_args()
    if a.selftest:
        sys.path.insert(0, ROOT)
        from cra5_amd import ops
        source = "cra5_amd (SELFTEST - not a known answer)"

        def enc(sym, idx, cdf, ln, off):
            return ops.rans_encode(sym, idx, cdf, ln, off)

        def dec(data, idx, cdf, ln, off):
            return [int(v) for v in ops.rans_decode(data, idx, cdf, ln, off)]
    else:
        import compressai
        from compressai import ans
        source = f"compressai {getattr(compressai, '__version__', '?')} compressai.ans"

        def _lists(cdf, ln, off):
            return [[int(v) for v in row] for row in cdf], [int(v) for v in ln], [int(v) for v in off]

        def enc(sym, idx, cdf, ln, off):
            c, l, o = _lists(cdf, ln, off)
            return ans.RansEncoder().encode_with_indexes([int(v) for v in sym], [int(v) for v in idx], c, l, o)

        def dec(data, idx, cdf, ln, off):
            c, l, o = _lists(cdf, ln, off)
            return ans.RansDecoder().decode_with_indexes(data, [int(v) for v in idx], c, l, o)
    store = dict(source=np.array(source), names=np.array([c[0] for c in cases()]))
    for name, sym, idx, cdf, ln, off in cases():
        data = enc(sym, idx, cdf, ln, off)
        back = np.asarray(dec(data, idx, cdf, ln, off), dtype=np.int32)
        assert np.array_equal(back, sym), f"{name}: the coder does not round-trip its own stream"
        store[f"stream_{name}"] = np.frombuffer(data, dtype=np.uint8)
        print(f"{name}: {sym.size} symbols -> {len(data)} bytes")
    np.savez_compressed(a.out, **store)
    print("wrote", a.out, "from", source)


if __name__ == "__main__":
    main()
