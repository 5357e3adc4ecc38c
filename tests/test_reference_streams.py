"""The REFERENCE's own integers through the PRODUCT's host coder (C ABI, no GPU needed).

`full268_ints.npz` / `thin_e2e.npz` hold every z symbol, CDF index and y symbol of frames the reference's Python ran
(tests/golden/make_golden.py --stage full / thin, synthetic weights), next to the byte strings (or their sha256) the
reference's `compress()` wrote for them (entropy_models.py:263-271, rans_interface.cpp:108-200).  Feeding those integers
to `cra5_rans_encode_with_indexes` with the product's own CDF tables must reproduce the reference-python-written streams byte for
byte - at the full 2.65 M-latent size too, where the end-to-end product run differs from the reference in a handful of
rounding flips (tests/test_model_gpu.py counts them) and a whole-stream comparison would otherwise never apply.
"""
import hashlib
import json

import numpy as np
import pytest

from cra5_amd import synth
from cra5_amd.entropy import EntropyBottleneck, GaussianConditional, get_scale_table
from oracle import cbind


def _gc():
    gc = GaussianConditional(None)
    assert gc.update_scale_table(get_scale_table(), force=True)
    return gc


def _eb(golden_dir, which, channels, seed):
    """The EntropyBottleneck of model `which` with the synthetic parameters the golden run used (per-key
    deterministic: cra5_amd/synth.py), tables built by the product's update()."""
    keys = json.load(open(f"{golden_dir}/state_keys.json"))[which]
    shapes = {k: tuple(v) for k, v in keys.items() if k.startswith("entropy_bottleneck.")}
    sd = synth.fill_state_dict(shapes, seed)
    eb = EntropyBottleneck(channels)
    own = eb.state_dict()
    for k, v in sd.items():
        own[k[len("entropy_bottleneck."):]].copy_(v)
    eb.update(force=True)
    return eb


def test_reference_integers_full_size_through_the_product_coder(golden_dir):
    g = np.load(f"{golden_dir}/full268_ints.npz")
    idx, sym = g["idx_full"].astype(np.int32), g["sym_full"].astype(np.int32)
    assert idx.size == sym.size == 256 * 72 * 144
    gc = _gc()
    y = gc.encode_symbols(sym, idx)
    assert len(y) == int(g["y_string_len"][0])
    assert hashlib.sha256(y).digest() == g["y_string_sha256"].tobytes()
    # ... the decoder (bucket tables, escape path: ~half of these symbols escape) reads them back, and so does the oracle's
    assert np.array_equal(gc.decode_symbols(y, idx), sym)
    cdf, ln, off = gc.host_tables()
    assert np.array_equal(np.asarray(cbind.rans_decode(y, idx, cdf, ln, off)).reshape(-1), sym)
    # z stream: 256 channels x 18 x 36 symbols, one CDF row per channel
    eb = _eb(golden_dir, "v268", 256, seed=7)
    zs = g["z_sym_full"].astype(np.int32)
    z_idx = eb._build_indexes((1, 256, 18, 36))
    z = eb.encode_symbols(zs, z_idx)
    assert z == g["z_string"].tobytes()
    assert np.array_equal(eb.decode_symbols(g["z_string"].tobytes(), z_idx), zs)


ROUND5_INTS = ["full268_m_ints", "bench1000_ints", "bench1000_m_ints", "bench1001_ints", "bench1001_m_ints"]


@pytest.mark.parametrize("which", ROUND5_INTS)
def test_reference_integers_round5_fixtures_through_the_product_coder(golden_dir, which):
    """Round 5 (tests/golden/make_golden.py --stage ints): the reference's integers of the 268 fixture frame under the
    entropy-matched weight variant (`_m`: sigma ~ rms(y), a trained model's regime - ~1 MB streams, a handful of
    escapes) and of the first two frames of the BENCHMARKED set (bench.py: synth_frame(268, 1000 + f)) under both
    variants.  Product coder on those integers == the reference-python-written streams (oracle coder plugged in as
    `compressai.ans`), and both decoders read them back."""
    g = np.load(f"{golden_dir}/{which}.npz")
    idx, sym = g["idx_full"].astype(np.int32), g["sym_full"].astype(np.int32)
    assert idx.size == sym.size == 256 * 72 * 144
    gc = _gc()
    y = gc.encode_symbols(sym, idx)
    assert len(y) == int(g["y_string_len"][0])
    assert hashlib.sha256(y).digest() == g["y_string_sha256"].tobytes()
    assert np.array_equal(gc.decode_symbols(y, idx), sym)
    cdf, ln, off = gc.host_tables()
    v = sym.astype(np.int64) - off[idx]
    n_esc = int(np.count_nonzero((v < 0) | (v >= ln[idx] - 2)))
    assert n_esc == int(g["n_escape"][0])
    if which.endswith("_m_ints"):
        assert n_esc < 0.01 * sym.size and len(y) < 1.5e6      # the matched variant's point: < 1 % escapes, ~1 MB
    eb = _eb(golden_dir, "v268", 256, seed=7)
    z_idx = eb._build_indexes((1, 256, 18, 36))
    assert eb.encode_symbols(g["z_sym_full"].astype(np.int32), z_idx) == g["z_string"].tobytes()


def test_reference_integers_thin_matched_through_the_product_coder(golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e_m.npz")
    idx, sym = g["idx_full"].astype(np.int32), g["sym_full"].astype(np.int32)
    gc = _gc()
    assert gc.encode_symbols(sym, idx) == g["y_string"].tobytes()
    assert np.array_equal(gc.decode_symbols(g["y_string"].tobytes(), idx), sym)
    eb = _eb(golden_dir, "thin", 16, seed=7)
    z_idx = eb._build_indexes((1, 16, 18, 36))
    assert eb.encode_symbols(g["z_sym_full"].astype(np.int32), z_idx) == g["z_string"].tobytes()


def test_reference_integers_thin_frame_a_through_the_product_coder(golden_dir):
    """Frame a of the thin model (round 5: input seed 135, stored in the fixture; rounds 1-4: seed 2, the one-index-flip
    frame): on the reference's integers the coder is byte-exact."""
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    idx, sym = g["idx_full"].astype(np.int32), g["sym_full"].astype(np.int32)
    gc = _gc()
    y = gc.encode_symbols(sym, idx)
    assert y == g["y_string"].tobytes()
    assert np.array_equal(gc.decode_symbols(g["y_string"].tobytes(), idx), sym)
    eb = _eb(golden_dir, "thin", 16, seed=7)
    z_idx = eb._build_indexes((1, 16, 18, 36))
    assert eb.encode_symbols(g["z_sym"].reshape(-1).astype(np.int32), z_idx) == g["z_string"].tobytes()


@pytest.mark.parametrize("which", ["full268_ints", "thin_e2e_b"])
def test_resolved_encoder_on_the_reference_integers(golden_dir, which):
    """The frame path encodes from device-resolved records (start | range, escape payload, nibble count); the host half
    of that route (`cra5_rans_encode_resolved`) on records resolved HERE from the reference's integers must write the
    same bytes as the one-call encoder (the device resolve kernel is held to the same records in test_kernels_gpu)."""
    from cra5_amd import ops
    g = np.load(f"{golden_dir}/{which}.npz")
    idx, sym = g["idx_full"].astype(np.int32), g["sym_full"].astype(np.int32)
    gc = _gc()
    cdf, ln, off = gc.host_tables()
    # host restatement of the device resolve kernel (csrc/elementwise.hip resolve_symbols_kernel; rans_interface.cpp
    # :120-160): value = sym - offset, escapes at both ends carry the folded payload and its nibble count
    max_v = (ln[idx] - 2).astype(np.int64)
    v = sym.astype(np.int64) - off[idx]
    neg, big = v < 0, v >= max_v
    raw = np.where(neg, -2 * v - 1, np.where(big, 2 * (v - max_v), 0)).astype(np.uint32)
    vc = np.where(neg | big, max_v, v)
    start = cdf[idx, vc].astype(np.uint32) & np.uint32(0xFFFF)
    rng = (cdf[idx, vc + 1] - cdf[idx, vc]).astype(np.uint32) & np.uint32(0xFFFF)
    sr = (start | (rng << np.uint32(16))).astype(np.uint32)
    nn = np.zeros(sym.size, np.int64)
    for k in range(8):      # nn = smallest n <= 8 with raw >> 4n == 0
        nn += (raw >> np.uint32(4 * k)) != 0
    nib = np.where(vc == max_v, nn + 1, 0).astype(np.uint8)
    y_res = ops.rans_encode_resolved(sr, raw, nib)
    y_one = gc.encode_symbols(sym, idx)
    assert y_res == y_one
