"""GPU parity of the whole path (thin-width full-spatial model and the real 268
architecture) against golden vectors produced by the REFERENCE's own Python
(tests/golden/make_golden.py) and against the CPU oracle, through the public
VAEformer / cra5_api surface (-> C ABI -> HIP kernels).

Tolerances (the reference's own fp32-vs-fp64 noise is y RMSE
5.5e-7, x_hat 9.4e-8 given identical y_hat, and 2 / 2.65M symbol flips end to end):
  * y (encode_to_latent)                       RMSE <= 1e-5
  * x_hat given an identical y_hat              RMSE <= 1e-5
  * integer side: index / symbol mismatches are counted and bounded (rounding flips of
    values that sit on a .5 boundary), never silently tolerated as float noise;
  * bitstreams: byte-identical for identical integer inputs.
"""
import hashlib

import numpy as np
import pytest
import torch

from cra5_amd import binfmt, ops, synth
from cra5_amd.vaeformer import VAEformer
from cra5_amd.zoo import vaeformer_pretrained
from oracle import cbind
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu


def rmse(a, b):
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    return float(torch.sqrt(torch.mean((a - b) ** 2)))


def sub(t, step):
    return t.detach().reshape(-1)[::step].cpu()


def synth_yhat(latent, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.round(2.0 * torch.randn(1, latent, 72, 144, generator=g)) + torch.randn(1, latent, 72, 144, generator=g)



def _inject_reference_zhat(net, z_sym_all, dev):
    """h_s of the PRODUCT on the REFERENCE's z_hat: the reference's integer z symbols (every one of them is in the
    fixture) + the medians = `quantize(z, "dequantize", medians)` of entropy_models.py:152-179.  Takes the product's own
    h_a / round() out of the comparison: means / scales / CDF indexes are then held to the float tolerance on EXACTLY the
    input the reference's h_s saw (vaeformer.py:350-376), whatever rounding flips the end-to-end run has."""
    cz = net.entropy_bottleneck.channels
    med = net.entropy_bottleneck.quantiles.detach()[:, 0, 1].reshape(cz, 1).float().to(dev)
    z_hat = torch.from_numpy(np.asarray(z_sym_all).reshape(cz, -1).astype(np.float32)).to(dev) + med
    scales, means = net._h_s_frame(z_hat.contiguous())
    return scales.contiguous(), means.contiguous()


def _explained_flips(name, got, ref, value, boundary_of, tol):
    """Element-wise integer comparison.  Every mismatch must be a step of exactly 1 AND sit within `tol` of the boundary
    it crossed (value = the product's float, boundary_of(lower integer) = the threshold between the two integers):
    a rounding flip is float noise on a discontinuous function, anything else is a bug.  Returns the flip count."""
    got, ref = np.asarray(got).reshape(-1).astype(np.int64), np.asarray(ref).reshape(-1).astype(np.int64)
    bad = np.nonzero(got != ref)[0]
    if bad.size:
        step = np.abs(got[bad] - ref[bad])
        assert step.max() == 1, f"{name}: {int((step > 1).sum())} mismatches by more than one step"
        v = np.asarray(value, dtype=np.float64).reshape(-1)[bad]
        b = boundary_of(np.minimum(got[bad], ref[bad]))
        d = np.abs(v - b) / np.maximum(1.0, np.abs(b))
        assert d.max() <= tol, f"{name}: a flipped element sits {d.max():.2e} from its boundary (tolerance {tol:.0e})"
    return int(bad.size)


@pytest.fixture(scope="module")
def thin(dev):
    net = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(net, seed=7)
    return net.to(dev)


@pytest.fixture(scope="module")
def thin_side(thin, dev, golden_dir):
    # frame a of the thin fixtures (thin_e2e.npz): the input seed travels with the fixture (round 5: 135, a frame on which
    # product and reference agree on every integer; rounds 1-4 used seed 2, the documented one-index-flip frame)
    x = synth.synth_frame(8, seed=int(np.load(f"{golden_dir}/thin_e2e.npz")["x_seed"][0])).unsqueeze(0).to(dev)
    y = thin.encode_latent(x, type='float')[0]
    s = thin._latent_side_frame(y[0], want_lik=True)
    torch.cuda.synchronize()
    return x, y, s


def test_thin_encoder_vs_reference_golden(thin_side, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    _, y, s = thin_side
    assert rmse(sub(y, 37), g["y_sub"]) <= 1e-5
    st = g["y_stats"]
    assert abs(float(y.double().sum()) - st[0]) <= 1e-5 * y.numel()
    # hyper-prior + entropy parameters
    z = s["z"].reshape(-1)
    assert rmse(z.cpu(), g["z"].reshape(-1)) <= 1e-5
    assert rmse(sub(s["scales"], 37), g["scales_sub"]) <= 1e-5
    assert rmse(sub(s["means"], 37), g["means_sub"]) <= 1e-5


def test_thin_integer_side_vs_reference_golden(thin_side, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    _, y, s = thin_side
    n = s["y_sym"].numel()
    z_mis = int((s["z_sym"].cpu().reshape(-1) != torch.from_numpy(g["z_sym"]).reshape(-1)).sum())
    idx_mis = int((sub(s["idx"], 37) != torch.from_numpy(g["idx_sub"])).sum())
    sym_mis = int((sub(s["y_sym"], 37) != torch.from_numpy(g["sym_sub"])).sum())
    hist = np.bincount((s["y_sym"].cpu().numpy().reshape(-1) + 256).clip(0, 512), minlength=513)
    hist_l1 = int(np.abs(hist - g["sym_hist"]).sum())
    idx_hist = np.bincount(s["idx"].cpu().numpy().reshape(-1), minlength=64)
    print(f"thin: z flips {z_mis}, idx flips {idx_mis}/{g['idx_sub'].size}, sym flips {sym_mis}/{g['sym_sub'].size}, "
          f"hist L1 {hist_l1}, idx hist L1 {int(np.abs(idx_hist - g['idx_hist']).sum())}")
    # a flip needs |value - boundary| < ~1e-6: expected rate ~1e-6 per element
    assert z_mis <= 2
    assert idx_mis <= 2 and sym_mis <= 2
    assert hist_l1 <= 2e-4 * n
    # likelihoods (forward() outputs): bits within 1e-4 relative
    bits_y = float((-torch.log2(s["y_lik"].double())).sum())
    bits_z = float((-torch.log2(s["z_lik"].double())).sum())
    assert abs(bits_y - g["bits_y"][0]) <= 2e-4 * g["bits_y"][0]
    assert abs(bits_z - g["bits_z"][0]) <= 2e-4 * g["bits_z"][0]


def test_thin_decoder_vs_reference_golden(thin, dev, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    x_hat = thin.decode_latent(synth_yhat(16, 5).to(dev))
    assert x_hat.shape == (1, 8, 721, 1440)
    assert rmse(sub(x_hat, 1009), g["xhat_sub"]) <= 1e-5
    assert rmse(x_hat[0, 0, 10].cpu(), g["xhat_row10_c0"]) <= 1e-5    # overlap row (two patches)
    assert rmse(x_hat[0, 0, 720].cpu(), g["xhat_row720_c0"]) <= 1e-5  # last row (i = 10 only)


def test_thin_roundtrip_and_bitstreams(thin, thin_side, dev, golden_dir, tmp_path, ledger):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    x, y, s = thin_side
    out = thin.compress_from_latent(y)
    assert tuple(out["z_shape"]) == (18, 36)
    y_str, z_str = out["strings"][0][0], out["strings"][1][0]
    # (1) stream equality vs the oracle coder for the same integer symbols + indexes
    eb, gc = thin.entropy_bottleneck, thin.gaussian_conditional
    o_y = cbind.rans_encode(s["y_sym"].cpu().numpy().reshape(-1), s["idx"].cpu().numpy().reshape(-1),
                            gc._quantized_cdf.cpu().numpy(), gc._cdf_length.cpu().numpy(), gc._offset.cpu().numpy())
    assert o_y == y_str
    # (2) decode(encode(sym)) == sym, y_hat reproduced bit-exactly on the decode side
    y_hat = thin.decompress(out["strings"], out["z_shape"], return_format='latent')
    assert torch.equal(y_hat[0].reshape(-1), s["y_hat"].reshape(-1))
    # (3) .bin container round trip, re-encode byte-equal
    blob = binfmt.pack_bin(out["strings"], out["z_shape"])
    strings2, shape2 = binfmt.unpack_bin(blob)
    assert strings2[0][0] == y_str and strings2[1][0] == z_str and shape2 == (18, 36)
    out2 = thin.compress_from_latent(y)
    assert out2["strings"][0][0] == y_str and out2["strings"][1][0] == z_str
    # (4) versus the stream the REFERENCE python produced (with the oracle coder).  Round 5: frame a is a frame on which the
    #     product agrees with the reference on EVERY integer (like frame b; rounds 1-4 used seed 2 with its one documented
    #     index flip and reported these comparisons as not applicable): plain asserts, element-wise first so that a
    #     regression names the integer that moved.
    assert torch.equal(s["z_sym"].cpu().reshape(-1), torch.from_numpy(g["z_sym"]).reshape(-1))
    assert z_str == g["z_string"].tobytes()
    ledger.ran("thin frame a: z stream == reference-python-written z stream")
    table = gc.scale_table.double().cpu().numpy()
    sc_np = np.maximum(s["scales"].double().cpu().numpy(), thin._scale_bound())
    n_iflip = _explained_flips("thin a CDF indexes", s["idx"].cpu().numpy(), g["idx_full"], sc_np,
                               lambda k: table[np.clip(k, 0, table.size - 1)], 1e-5)
    resid = (y[0].double() - s["means"].double()).cpu().numpy()
    n_sflip = _explained_flips("thin a y symbols", s["y_sym"].cpu().numpy(), g["sym_full"], resid, lambda k: k + 0.5, 1e-4)
    print(f"thin frame a vs the reference run: CDF index flips {n_iflip}, y symbol flips {n_sflip} of {g['idx_full'].size}")
    assert (n_iflip, n_sflip) == (0, 0)
    assert y_str == g["y_string"].tobytes()
    ledger.ran("thin frame a: y stream == reference-python-written y stream", f"{len(y_str)} bytes")
    # the reference's integers through the device resolve kernel + host state-update loop: the reference-python-written stream
    sr, raw, esc = ops.rans_resolve_symbols(torch.from_numpy(g["sym_full"].astype(np.int32)).to(dev),
                                            torch.from_numpy(g["idx_full"].astype(np.int32)).to(dev),
                                            gc._quantized_cdf, gc._cdf_length, gc._offset)
    assert ops.rans_encode_resolved(sr.cpu().numpy(), raw.cpu().numpy(), esc.cpu().numpy()) == g["y_string"].tobytes()
    ledger.ran("thin frame a: y stream coded from the reference's integers == reference-python-written y stream")
    # (5) full decode: x_hat from the stream == decode_latent(y_hat) exactly (deterministic kernels)
    xa = thin.decompress(out["strings"], out["z_shape"])["x_hat"]
    xb = thin.decode_latent(y_hat)
    assert torch.equal(xa, xb)
    assert torch.isfinite(xa).all()


# ---- second thin fixture frame (thin_e2e_b.npz, input seed 162): chosen by tests/golden/make_golden.py
# stage_thin_search + tools/thin_seed_probe.py so that product, oracle and reference agree on EVERY integer of the
# frame - the byte comparison with the reference-python-written stream and the cross-implementation decode are plain
# asserts here, no xfail, no `if`


@pytest.fixture(scope="module")
def thin_b(thin, dev, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e_b.npz")
    x = synth.synth_frame(8, seed=int(g["x_seed"][0])).unsqueeze(0).to(dev)
    y = thin.encode_latent(x, type='float')[0]
    s = thin._latent_side_frame(y[0], want_lik=True)
    torch.cuda.synchronize()
    return g, x, y, s


def test_thin_b_integer_side_identical_to_reference(thin_b):
    g, x, y, s = thin_b
    assert rmse(sub(y, 37), g["y_sub"]) <= 1e-5
    z_mis = int((s["z_sym"].cpu().reshape(-1) != torch.from_numpy(g["z_sym"]).reshape(-1)).sum())
    i_mis = int((s["idx"].cpu().reshape(-1).numpy() != g["idx_full"].astype(np.int32)).sum())
    y_mis = int((s["y_sym"].cpu().reshape(-1).numpy() != g["sym_full"].astype(np.int32)).sum())
    print(f"thin frame b: z flips {z_mis}, idx flips {i_mis}, y-symbol flips {y_mis} of {g['idx_full'].size} "
          f"(reference margins z {g['margin_z'][0]:.1e}, y {g['margin_y'][0]:.1e}, scale {g['margin_scale'][0]:.1e})")
    assert (z_mis, i_mis, y_mis) == (0, 0, 0)
    assert rmse(sub(s["means"], 37), g["means_sub"]) <= 1e-5 and rmse(sub(s["scales"], 37), g["scales_sub"]) <= 1e-5


def test_thin_b_streams_equal_the_reference_written_streams(thin, thin_b, ledger):
    """compress() of the product == the bytes the REFERENCE's compress() wrote for the same frame (through the oracle
    coder as `compressai.ans`), both streams, byte for byte."""
    g, x, y, s = thin_b
    out = thin.compress(x)
    assert tuple(out["z_shape"]) == (18, 36)
    assert out["strings"][1][0] == g["z_string"].tobytes()
    assert hashlib.sha256(out["strings"][0][0]).digest() == g["y_string_sha256"].tobytes()
    assert out["strings"][0][0] == g["y_string"].tobytes()
    ledger.ran("thin frame b (seed 162): y and z streams == reference-python-written streams",
               f"{len(out['strings'][0][0])} + {len(out['strings'][1][0])} bytes")


def test_thin_b_decodes_the_reference_stream(thin, thin_b, ledger):
    """Cross-implementation decode: the strings the REFERENCE wrote, through the product's decompress(): the
    reference's y_hat and the reference's own reconstruction from its own stream."""
    g, x, y, s = thin_b
    strings = [[g["y_string"].tobytes()], [g["z_string"].tobytes()]]
    y_hat = thin.decompress(strings, (18, 36), return_format='latent')
    d = (sub(y_hat, 37) - torch.from_numpy(g["y_hat_sub"])).abs()
    assert float(d.max()) <= 1e-4, float(d.max())     # same symbols + means within fp32 noise
    sym = torch.round(y_hat[0].reshape(-1) - s["means"].reshape(-1)).int().cpu().numpy()
    assert np.array_equal(sym, g["sym_full"].astype(np.int32))
    x_hat = thin.decompress(strings, (18, 36))["x_hat"]
    e = rmse(sub(x_hat, 1009), g["xhat_rt_sub"])
    print(f"thin frame b: x_hat decoded from the reference's stream vs the reference's own decode: rmse {e:.2e}")
    assert e <= 1e-5
    ledger.ran("thin frame b (seed 162): reference-python-written .bin strings decode to the reference's y_hat / x_hat",
               f"x_hat rmse {e:.1e}")


def test_thin_decodes_the_reference_stream(thin, thin_side, dev, golden_dir, ledger):
    """Cross-implementation decode: the `.bin` strings the REFERENCE's compress() wrote (thin_e2e.npz,
    through the oracle coder) fed to the product's decompress().  Needs bit-identical CDF indexes from
    the product's h_s; a single flipped index desynchronises the rest of the stream,
    so the integer side is checked first and a mismatch is reported as such, never skipped silently."""
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    _, y, s = thin_side
    assert torch.equal(s["z_sym"].cpu().reshape(-1), torch.from_numpy(g["z_sym"]).reshape(-1))
    idx_ref = g["idx_full"].astype(np.int32)
    assert np.array_equal(s["idx"].cpu().numpy().reshape(-1), idx_ref)        # element-wise, not a histogram
    strings = [[g["y_string"].tobytes()], [g["z_string"].tobytes()]]
    y_hat = thin.decompress(strings, (18, 36), return_format='latent')
    ledger.ran("thin frame a: reference-python-written stream decodes on this build")
    # ... and the decoder alone on the reference's CDF indexes (what rounds 1-4 had to fall back to on the flip frame)
    sym = thin.gaussian_conditional.decode_symbols(strings[0][0], idx_ref)
    assert np.array_equal(sym, g["sym_full"].astype(np.int32))
    d = (sub(y_hat, 37) - torch.from_numpy(g["y_hat_sub"])).abs()
    assert float(d.max()) <= 1e-4, float(d.max())     # same symbols + means within fp32 noise
    sym_dec = torch.round(y_hat[0].reshape(-1) - s["means"].reshape(-1)).int().cpu().numpy()
    assert np.array_equal(sym_dec, g["sym_full"].astype(np.int32))


# ---- the documented FLIP frame (thin_e2e_flip.npz, input seed 2: rounds 1-4's frame a; round 6) ---------------
# Frames a (seed 135) and b (seed 162) were SELECTED for agreement (tools/thin_seed_probe.py: 20 of 36 probed thin frames
# agree with the reference run on every integer).  This one is kept because it does not: the product's h_s puts ONE of the
# 165 888 scales on the other side of a scale-table entry (within 1e-5 of it).  Pinned here: how many integers differ
# and how far from their boundary, that the reference-written y stream is REFUSED (round 6: the decoder's end-state
# check; the reference decodes garbage silently), and that the decoder reads it with the reference's indexes injected.


@pytest.fixture(scope="module")
def thin_flip(thin, dev, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e_flip.npz")
    x = synth.synth_frame(8, seed=int(g["x_seed"][0])).unsqueeze(0).to(dev)
    y = thin.encode_latent(x, type='float')[0]
    s = thin._latent_side_frame(y[0], want_lik=True)
    torch.cuda.synchronize()
    return g, x, y, s


def test_thin_flip_frame_pinned(thin, thin_flip, dev, ledger):
    from cra5_amd._lib import Cra5Error
    g, x, y, s = thin_flip
    gc = thin.gaussian_conditional
    assert rmse(sub(y, 37), g["y_sub"]) <= 1e-5
    assert rmse(sub(s["scales"], 37), g["scales_sub"]) <= 1e-5 and rmse(sub(s["means"], 37), g["means_sub"]) <= 1e-5
    # z symbols: all equal (a regression here fails) -> the z stream is the reference-written one
    assert torch.equal(s["z_sym"].cpu().reshape(-1), torch.from_numpy(g["z_sym"]).reshape(-1))
    out = thin.compress_from_latent(y)
    y_str, z_str = out["strings"][0][0], out["strings"][1][0]
    assert z_str == g["z_string"].tobytes()
    table = gc.scale_table.double().cpu().numpy()
    sc_np = np.maximum(s["scales"].double().cpu().numpy(), thin._scale_bound())
    n_iflip = _explained_flips("thin flip-frame CDF indexes", s["idx"].cpu().numpy(), g["idx_full"], sc_np,
                               lambda k: table[np.clip(k, 0, table.size - 1)], 1e-5)
    resid = (y[0].double() - s["means"].double()).cpu().numpy()
    n_sflip = _explained_flips("thin flip-frame y symbols", s["y_sym"].cpu().numpy(), g["sym_full"], resid,
                               lambda k: k + 0.5, 1e-4)
    print(f"thin flip frame (seed 2) vs the reference run: CDF index flips {n_iflip}, y symbol flips {n_sflip} of "
          f"{g['idx_full'].size}")
    assert n_iflip <= 1 and n_sflip == 0        # pinned: rounds 3-5 measured exactly (1, 0); more is a regression
    ref_strings = [[g["y_string"].tobytes()], [g["z_string"].tobytes()]]
    if n_iflip == 0:
        assert y_str == g["y_string"].tobytes()
        thin.decompress(ref_strings, (18, 36), return_format='latent')
        ledger.ran("thin flip frame (seed 2): y stream == reference-python-written y stream (no flip on this build)")
    else:
        assert abs(len(y_str) - int(g["y_string_len"][0])) <= 64
        ledger.not_applicable("thin flip frame (seed 2): y stream == reference-python-written y stream",
                              f"documented flip case, kept on purpose: {n_iflip} of {g['idx_full'].size} CDF indexes one step "
                              "off, scale within 1e-5 of the table entry (frames a / b were selected for agreement; "
                              "20 of 36 probed thin frames agree on every integer)")
        # the reference-written stream desynchronises at that element: an ERROR here (the reference's own decoder
        # would hand back a wrong frame), for the latent and for the reconstruction route
        with pytest.raises(Cra5Error) as ei:
            thin.decompress(ref_strings, (18, 36), return_format='latent')
        assert "y stream" in str(ei.value) or ei.value.status == -6
        with pytest.raises(Cra5Error):
            thin.decompress(ref_strings, (18, 36))
        ledger.ran("thin flip frame (seed 2): reference-python-written stream with one foreign CDF index is REFUSED",
                   type(ei.value).__name__)
    # the product's own stream of this frame decodes (its own indexes), bit-exactly
    y_hat = thin.decompress(out["strings"], out["z_shape"], return_format='latent')
    assert torch.equal(y_hat[0].reshape(-1), s["y_hat"].reshape(-1))
    # the decoder itself, held to the reference-written stream with the REFERENCE's indexes injected
    idx_ref = g["idx_full"].astype(np.int32)
    sym = gc.decode_symbols(ref_strings[0][0], idx_ref)
    assert np.array_equal(sym, g["sym_full"].astype(np.int32))
    ledger.ran("thin flip frame (seed 2): reference-python-written stream decodes with the reference's CDF indexes injected")
    # ... and the reference's integers through the product coder write the reference's bytes
    sr, raw, esc = ops.rans_resolve_symbols(torch.from_numpy(g["sym_full"].astype(np.int32)).to(dev),
                                            torch.from_numpy(idx_ref).to(dev), gc._quantized_cdf, gc._cdf_length, gc._offset)
    assert ops.rans_encode_resolved(sr.cpu().numpy(), raw.cpu().numpy(), esc.cpu().numpy()) == g["y_string"].tobytes()


def test_reference_written_streams_decode_or_are_refused(thin, dev, golden_dir, ledger):
    """VERDICT r5 item 4, the honest form of "drop-in decoder".  For each of the 36 probed thin frames (thin_cands.npz: the
    REFERENCE's integers; the streams its compress() writes are rebuilt from them - coder on the reference's integers ==
    reference-python-written bytes, tests/test_reference_streams.py) the product's decompress() must EITHER return exactly
    the reference's symbols (every CDF index this build derives equals the encoder's) OR raise - never return a frame
    decoded past a foreign index.  rans_interface.cpp:215-284 has no such check."""
    from cra5_amd._lib import Cra5Error, StreamDesyncError
    c = np.load(f"{golden_dir}/thin_cands.npz")
    eb, gc = thin.entropy_bottleneck, thin.gaussian_conditional
    z_idx = eb._build_indexes((1, eb.channels, 18, 36))
    decoded = refused = desync = benign = 0
    for k, seed in enumerate(c["seeds"]):
        z_ref = c["z_sym"][k].astype(np.int32)
        idx_ref, sym_ref = c["idx_full"][k].astype(np.int32), c["sym_full"][k].astype(np.int32)
        z_str = eb.encode_symbols(z_ref, z_idx)
        y_str = gc.encode_symbols(sym_ref, idx_ref)
        # what THIS build derives from the reference's z stream
        sc, mu = _inject_reference_zhat(thin, z_ref, dev)
        idx_own = ops.gaussian_conditional(sc, mu, gc.scale_table, sym_in=torch.zeros_like(mu, dtype=torch.int32),
                                           want=("idx",), scale_bound=thin._scale_bound())["idx"].cpu().numpy().reshape(-1)
        agree = np.array_equal(idx_own, idx_ref)
        try:
            y_hat = thin.decompress([[y_str], [z_str]], (18, 36), return_format='latent')
        except Cra5Error as e:
            assert not agree, f"seed {seed}: every CDF index agrees with the encoder's, yet the stream was refused: {e}"
            refused += 1
            desync += isinstance(e, StreamDesyncError)
            continue
        # it decoded: then it decoded the REFERENCE's symbols, all of them - never a frame read past a foreign index.  (A
        # foreign index can be harmless: neighbouring table rows share the (start, range) of bins in their tails; the
        # coder then stays in step and the end-state check passes - that is a correct decode, not a missed one.)
        sym = torch.round(y_hat[0].reshape(-1) - mu.reshape(-1)).int().cpu().numpy()
        assert np.array_equal(sym, sym_ref), f"seed {seed}: decoded without an error but not to the reference's symbols"
        decoded += 1
        benign += not agree
    print(f"{len(c['seeds'])} reference-written thin frames: {decoded} decode to the reference's symbols ({benign} of them "
          f"across a harmless foreign index), {refused} refused ({desync} by the end-state check, {refused - desync} by a "
          "stream error on the way)")
    assert decoded + refused == len(c["seeds"]) and decoded >= 12 and refused >= 6     # round 5 probe: 20 agree on every index
    ledger.ran("36 reference-written thin frames: decode exactly or are refused, never garbage",
               f"{decoded} decoded ({benign} across a harmless foreign index), {refused} refused ({desync} CRA5_ERR_DESYNC)")


def test_thin_vs_cpu_oracle(thin, thin_side, dev):
    """Same seeded weights/input through oracle/torch_ref.py on the host cores."""
    x, y, s = thin_side
    cfg = R.cfg_thin()
    sd = {k: v.detach().cpu() for k, v in thin.state_dict().items()}
    y_ref = R.encode_y(x.cpu(), sd, cfg)
    assert rmse(y, y_ref) <= 1e-5
    side = R.latent_side(y.cpu(), sd, cfg, thin.gaussian_conditional.scale_table.cpu())
    assert rmse(s["means"], side["means"]) <= 1e-5 and rmse(s["scales"], side["scales"]) <= 1e-5
    assert int((s["idx"].cpu().reshape(-1) != side["idx"].reshape(-1)).sum()) <= 2
    assert int((s["y_sym"].cpu().reshape(-1) != side["y_sym"].reshape(-1)).sum()) <= 2
    fw = thin(x)
    assert fw["x_hat"].shape == x.shape and fw["likelihoods"]["y"].shape == y.shape
    assert fw["likelihoods"]["z"].shape == (1, 16, 18, 36)
    x_ref = R.g_s(s["y_hat"].reshape(1, 16, 72, 144).cpu(), sd, cfg)
    assert rmse(fw["x_hat"], x_ref) <= 1e-5


def synth_zhat(cz, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.round(3.0 * torch.randn(1, cz, 18, 36, generator=g))


def _hs_from_synth(net, dev):
    """Hyper-decoder on a regenerable synthetic z_hat: independent of rounding flips of z."""
    cz = net.entropy_bottleneck.channels
    zs = synth_zhat(cz, 6)[0].reshape(cz, -1).to(dev) + net.entropy_bottleneck.quantiles[:, 0, 1].reshape(cz, 1)
    scales, means = net._h_s_frame(zs.contiguous())
    return torch.cat([scales, means], 0)


def test_thin_hyper_decoder_vs_reference_golden(thin, dev, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    p = _hs_from_synth(thin, dev)
    assert rmse(sub(p, 37), g["hs_synth_sub"]) <= 1e-5
    idx = ops.gaussian_conditional(p[:16].contiguous(), p[16:].contiguous(), thin.gaussian_conditional.scale_table,
                                   sym_in=torch.zeros(16, 72, 144, dtype=torch.int32, device=dev), want=("idx",))["idx"]
    hist = np.bincount(idx.cpu().numpy().reshape(-1), minlength=64)
    assert np.abs(hist - g["hs_synth_idx_hist"]).sum() <= 4


def test_thin_determinism(thin, thin_side, dev):
    """h_s must give bit-identical indexes / means on repeated runs (encode vs decode side)."""
    _, y, s = thin_side
    s2 = thin._latent_side_frame(y[0])
    assert torch.equal(s2["idx"], s["idx"]) and torch.equal(s2["means"], s["means"])
    assert torch.equal(s2["y_sym"], s["y_sym"])


def test_api_surface_thin(thin, dev, tmp_path):
    """cra5_api methods with an in-memory frame and synthetic weights (268-channel stats
    do not apply to the 8-channel thin model: use unit statistics)."""
    from cra5_amd.api import cra5_api
    api = cra5_api(local_root=str(tmp_path), device="cuda", weights=thin)
    api._mean_flat = torch.linspace(-1, 1, 8, device=dev)
    api._std_flat = torch.linspace(0.5, 2, 8, device=dev)
    api.mean, api.std = api._mean_flat.view(8, 1, 1), api._std_flat.view(8, 1, 1)
    frame = synth.synth_frame(8, seed=3) * api.std.cpu() + api.mean.cpu()
    y = api.encode_to_latent("2024-06-01T00:00:00", data=frame)
    assert y.shape == (1, 16, 72, 144)
    # fused normalisation == explicit normalisation
    y2 = thin.encode_latent(api.normalization(frame.to(dev)).unsqueeze(0), type='float')[0]
    assert rmse(y, y2) <= 1e-6
    b = api.latent_to_bin(y)
    assert set(b) == {"strings", "z_shape"} and len(b["strings"]) == 2
    r = api.encode_era5_as_bin("2024-06-01T00:00:00", save_root=str(tmp_path / "CRA5"), data=frame)
    assert set(r) == {"output", "reading_time", "encoding_time", "saving_time", "save_path"}
    assert r["save_path"].endswith("CRA5/2024/2024-06-01T00:00:00.bin")
    y_hat = api.bin_to_latent(r["save_path"])
    x_norm = api.latent_to_reconstruction(y_hat)
    assert x_norm.shape == (1, 8, 721, 1440)
    d = api.decode_from_bin("2024-06-01T00:00:00", return_format='normalized')
    assert torch.equal(d["x_hat"], x_norm)
    d2 = api.decode_from_bin("2024-06-01T00:00:00", return_format='de_normalized')
    ref = x_norm[0] * api.std + api.mean
    assert rmse(d2["x_hat"], ref) <= 1e-5
    assert api.decode_from_bin("2024-06-01T00:00:00", return_format='latent').shape == (1, 16, 72, 144)
    # host arrays in / out through the chunked pinned staging (csrc/runtime.hip): same bits as the device path
    fr_np = frame.numpy()
    assert torch.equal(api._frame(None, fr_np).cpu(), frame)
    assert torch.equal(api.encode_to_latent("2024-06-01T00:00:00", data=fr_np), y)
    d3 = api.decode_from_bin("2024-06-01T00:00:00", return_format='de_normalized', to_host=True)
    assert isinstance(d3["x_hat"], np.ndarray) and np.array_equal(d3["x_hat"], d2["x_hat"].reshape(8, 721, 1440).cpu().numpy())
    buf = np.zeros((8, 721, 1440), dtype=np.float32)
    assert api.decode_from_bin("2024-06-01T00:00:00", return_format='normalized', out=buf)["x_hat"] is buf
    assert np.array_equal(buf, x_norm[0].cpu().numpy())
    with pytest.raises(ValueError, match="C-contiguous float32"):
        api.decode_from_bin("2024-06-01T00:00:00", out=np.zeros((8, 721, 1439), dtype=np.float32))
    # a frame with a masked (NaN) value is refused at the encode edge instead of being coded as garbage
    bad = fr_np.copy()
    bad[3, 100, 200] = np.nan
    with pytest.raises(ValueError, match="NaN / inf"):
        api.encode_era5_as_bin("2024-06-01T00:00:00", save_root=str(tmp_path / "CRA5"), data=bad)
    with pytest.raises(ValueError, match="NaN / inf"):
        api.encode_to_latent("2024-06-01T00:00:00", data=bad)


# --------------------------------------------------------------------------------------
# the real architecture (268 variables, 404.7 M parameters)
# --------------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def big(dev):
    net = vaeformer_pretrained(quality=268, pretrained=False)
    synth.load_synthetic(net, seed=7)
    return net.to(dev)


def test_full268_vs_reference_golden(big, dev, golden_dir, ledger):
    g = np.load(f"{golden_dir}/full268.npz")
    x = synth.synth_frame(268, seed=2).unsqueeze(0).to(dev)
    y = big.encode_latent(x, type='float')[0]
    e_y = rmse(sub(y, 499), g["y_sub"])
    s = big._latent_side_frame(y[0], want_lik=True)
    # z is quantised before h_s: ONE flipped z symbol (a value within ~1e-5 of a .5 boundary)
    # moves every mean / scale by ~1e-3 through the global attention of h_s.  So: count the z
    # flips, compare means / scales directly only when there are none, and pin h_s itself on
    # a regenerable synthetic z_hat (next block).
    z_sub = s["z_sym"].cpu().reshape(-1)[::13]
    z_flips = int((z_sub != torch.from_numpy(g["z_sym"]).reshape(-1)).sum())
    z_hist = np.bincount((s["z_sym"].cpu().numpy().reshape(-1) + 64).clip(0, 128), minlength=129)
    z_hist_l1 = int(np.abs(z_hist - g["z_sym_hist"]).sum())
    e_z = rmse(sub(s["z"], 13), g["z"])
    e_m, e_s = rmse(sub(s["means"], 499), g["means_sub"]), rmse(sub(s["scales"], 499), g["scales_sub"])
    idx_mis = int((sub(s["idx"], 499) != torch.from_numpy(g["idx_sub"])).sum())
    sym_mis = int((sub(s["y_sym"], 499) != torch.from_numpy(g["sym_sub"])).sum())
    hist = np.bincount((s["y_sym"].cpu().numpy().reshape(-1) + 256).clip(0, 512), minlength=513)
    idx_hist = np.bincount(s["idx"].cpu().numpy().reshape(-1), minlength=64)
    print(f"268: y rmse {e_y:.3e}, z rmse {e_z:.3e}, z flips (1/13 sample) {z_flips}, z hist L1 {z_hist_l1}, "
          f"means {e_m:.3e}, scales {e_s:.3e}, idx flips {idx_mis}, sym flips {sym_mis}, "
          f"sym hist L1 {int(np.abs(hist - g['sym_hist']).sum())}, idx hist L1 {int(np.abs(idx_hist - g['idx_hist']).sum())}")
    assert e_y <= 1e-5
    # forward()'s likelihood outputs (vaeformer.py:302-333) at full size: total bits within 2e-4 of the reference's
    bits_y = float((-torch.log2(s["y_lik"].double())).sum())
    bits_z = float((-torch.log2(s["z_lik"].double())).sum())
    print(f"268: bits y {bits_y:.6e} (ref {g['bits_y'][0]:.6e}), bits z {bits_z:.6e} (ref {g['bits_z'][0]:.6e})")
    assert abs(bits_z - g["bits_z"][0]) <= 2e-4 * g["bits_z"][0]
    z_rms = float(np.sqrt(g["z_stats"][1] / s["z"].numel()))
    assert e_z <= 1e-5 * max(1.0, z_rms)   # z is O(7) with the synthetic gains: relative 1e-5
    # z mod 1 is ~uniform, so a symbol flips with probability 2|err|: expect numel * 2 * 0.8 * rmse
    # flips (4.8 for the reference's own fp32-vs-fp64 error of 1.8e-5); each moves 2 histogram counts.
    expected_flips = s["z"].numel() * 2 * 0.8 * e_z
    assert z_hist_l1 <= max(8, 2 * 3 * expected_flips), (z_hist_l1, expected_flips)
    # End to end (the product's own z symbols): a flipped z symbol moves every mean / scale by ~1e-3 through h_s, so
    # means / scales of THIS run are only compared when no symbol flipped; the float comparison proper runs below on
    # the reference's z_hat, un-conditionally.
    gi = np.load(f"{golden_dir}/full268_ints.npz")
    z_all = s["z_sym"].cpu().numpy().reshape(-1)
    med_np = big.entropy_bottleneck.quantiles.detach()[:, 0, 1].cpu().numpy().astype(np.float64)
    z_val = s["z"].double().cpu().numpy().reshape(256, -1) - med_np[:, None]
    n_zflip = _explained_flips("268 z symbols", z_all, gi["z_sym_full"], z_val, lambda k: k + 0.5, 1e-4)
    assert z_hist_l1 <= 2 * n_zflip
    print(f"268: z symbol flips end to end {n_zflip} of {z_all.size}, each one step across a .5 boundary within 1e-4")
    if n_zflip == 0:
        assert e_m <= 1e-5 and e_s <= 1e-5
    # ---- the reference's z_hat injected into the product's h_s --------------------------
    sc_i, mu_i = _inject_reference_zhat(big, gi["z_sym_full"], dev)
    e_mi, e_si = rmse(sub(mu_i, 499), g["means_sub"]), rmse(sub(sc_i, 499), g["scales_sub"])
    print(f"268, reference z_hat injected: means rmse {e_mi:.3e}, scales rmse {e_si:.3e}")
    assert e_mi <= 1e-5 and e_si <= 1e-5
    ledger.ran("268 full size: means / scales <= 1e-5 on the reference's z_hat (injected into the product's h_s)",
               f"means {e_mi:.1e}, scales {e_si:.1e}")
    gci = ops.gaussian_conditional(sc_i, mu_i, big.gaussian_conditional.scale_table, y=y[0].contiguous(),
                                   want=("idx", "sym"), scale_bound=big._scale_bound(),
                                   lik_bound=big.gaussian_conditional.likelihood_bound)
    table = big.gaussian_conditional.scale_table.double().cpu().numpy()
    sc_np = np.maximum(sc_i.double().cpu().numpy(), big._scale_bound())
    n_iflip = _explained_flips("268 CDF indexes", gci["idx"].cpu().numpy(), gi["idx_full"], sc_np,
                               lambda k: table[np.clip(k, 0, table.size - 1)], 1e-5)
    resid = (y[0].double() - mu_i.double()).cpu().numpy()
    n_sflip = _explained_flips("268 y symbols", gci["sym"].cpu().numpy(), gi["sym_full"], resid, lambda k: k + 0.5, 1e-4)
    n_lat = gi["idx_full"].size
    print(f"268, reference z_hat injected: CDF index flips {n_iflip} / {n_lat} (reference vs its own fp64 run: 8), "
          f"y symbol flips {n_sflip} / {n_lat} (expected from the y error: {n_lat * 2 * 0.8 * e_y:.0f})")
    assert n_iflip <= 16                                   # scale within 1e-5 of a table entry: O(10) of 2.65 M
    assert n_sflip <= max(8, 3 * n_lat * 2 * 0.8 * e_y)    # |y - mu| within the y error of a .5 boundary
    ledger.ran("268 full size: every CDF index / y symbol vs the reference's, flips explained element-wise",
               f"{n_iflip} index, {n_sflip} symbol flips of {n_lat}, all one step across a boundary within tolerance")
    # the reference's integers through the DEVICE resolve kernel + host state-update loop (the frame path's coder):
    # the stream the reference's compress() wrote, byte for byte (sha256; the CPU twin is tests/test_reference_streams.py)
    gcm = big.gaussian_conditional
    sr, raw, esc = ops.rans_resolve_symbols(torch.from_numpy(gi["sym_full"].astype(np.int32)).to(dev),
                                            torch.from_numpy(gi["idx_full"].astype(np.int32)).to(dev),
                                            gcm._quantized_cdf, gcm._cdf_length, gcm._offset)
    y_ref_stream = ops.rans_encode_resolved(sr.cpu().numpy(), raw.cpu().numpy(), esc.cpu().numpy())
    assert len(y_ref_stream) == int(gi["y_string_len"][0])
    assert hashlib.sha256(y_ref_stream).digest() == gi["y_string_sha256"].tobytes()
    ledger.ran("268 full size: sha256(y stream coded from the reference's integers) == reference-python-written y stream's",
               f"{len(y_ref_stream)} bytes through the device resolve kernel + cra5_rans_encode_resolved")
    z_idx_all = big.entropy_bottleneck._build_indexes((1, 256, 18, 36))
    assert big.entropy_bottleneck.encode_symbols(gi["z_sym_full"].astype(np.int32), z_idx_all) == g["z_string"].tobytes()
    ledger.ran("268 full size: z stream coded from the reference's z symbols == reference-python-written z stream",
               f"{g['z_string'].size} bytes")
    p = _hs_from_synth(big, dev)
    e_h = rmse(sub(p, 499), g["hs_synth_sub"])
    print(f"268: h_s on synthetic z_hat rmse {e_h:.3e}")
    assert e_h <= 1e-5
    # x -> bin -> y_hat round trip at full size
    out = big.compress_from_latent(y)
    assert abs(len(out["strings"][0][0]) - int(g["y_string_len"][0])) <= 256
    # full-size, escape-heavy streams byte-for-byte against the oracle coder on the same integers
    # (entropy_models.py:263-271 -> rans_interface.cpp:108-200)
    eb, gc = big.entropy_bottleneck, big.gaussian_conditional
    o_y = cbind.rans_encode(s["y_sym"].cpu().numpy().reshape(-1), s["idx"].cpu().numpy().reshape(-1),
                            gc._quantized_cdf.cpu().numpy(), gc._cdf_length.cpu().numpy(), gc._offset.cpu().numpy())
    assert out["strings"][0][0] == o_y
    z_idx = eb._build_indexes((1,) + tuple(s["z_sym"].shape)).reshape(-1)
    o_z = cbind.rans_encode(s["z_sym"].cpu().numpy().reshape(-1), z_idx, eb._quantized_cdf.cpu().numpy(),
                            eb._cdf_length.cpu().numpy(), eb._offset.cpu().numpy())
    assert out["strings"][1][0] == o_z
    # ... and the oracle DECODER reads the product's stream back to the same symbols
    back = cbind.rans_decode(out["strings"][0][0], s["idx"].cpu().numpy().reshape(-1), gc._quantized_cdf.cpu().numpy(),
                             gc._cdf_length.cpu().numpy(), gc._offset.cpu().numpy())
    assert torch.equal(back.reshape(-1), s["y_sym"].cpu().reshape(-1))
    sym_l1, idx_l1 = int(np.abs(hist - g["sym_hist"]).sum()), int(np.abs(idx_hist - g["idx_hist"]).sum())
    # the product's OWN end-to-end streams equal the reference-python-written ones exactly when no integer flipped (the
    # comparison on the reference's integers ran above, un-conditionally)
    if n_zflip == 0:
        assert out["strings"][1][0] == g["z_string"].tobytes()
    if n_zflip == 0 and sym_l1 == 0 and idx_l1 == 0:
        assert hashlib.sha256(out["strings"][0][0]).digest() == g["y_string_sha256"].tobytes()
    print(f"268 end to end: z flips {n_zflip}, symbol histogram L1 {sym_l1}, index histogram L1 {idx_l1} "
          "(the reference flips 2 symbols / 8 indexes against its own fp64 run, BASELINE.md section 2)")
    y_hat = big.decompress(out["strings"], out["z_shape"], return_format='latent')
    assert torch.equal(y_hat[0].reshape(-1), s["y_hat"].reshape(-1))
    # decoder, identical y_hat
    x_hat = big.decode_latent(synth_yhat(256, 5).to(dev))
    e_x = rmse(sub(x_hat, 99991), g["xhat_sub"])
    print(f"268: x_hat rmse {e_x:.3e} (given identical y_hat)")
    assert e_x <= 1e-5
    assert rmse(x_hat[0, 0, 10].cpu(), g["xhat_row10_c0"]) <= 1e-5
    assert rmse(x_hat[0, 0, 720].cpu(), g["xhat_row720_c0"]) <= 1e-5


@pytest.mark.parametrize("which,variant", [("full268_m_ints", "matched"), ("bench1000_ints", "default"),
                                           ("bench1000_m_ints", "matched"), ("bench1001_ints", "default"),
                                           ("bench1001_m_ints", "matched")])
def test_full268_round5_reference_integers(big, dev, golden_dir, ledger, which, variant):
    """Round 5 fixtures (make_golden.py --stage ints): the reference run on the 268 fixture frame under the
    entropy-matched weight variant, and on the first two frames of the BENCHMARKED set (bench.py codes exactly these
    tensors: synth_frame(268, 1000 + f)) under both variants.  Same ladder as test_full268_vs_reference_golden: y within
    1e-5; the reference's z_hat injected into the product's h_s -> mu / sigma within 1e-5, every CDF index and y symbol
    element-wise with every flip explained; the reference's integers through the device resolve kernel + host coder ==
    the reference-python-written streams (sha256)."""
    gi = np.load(f"{golden_dir}/{which}.npz")
    seed = int(gi["x_seed"][0])
    synth.apply_variant(big, seed=7, variant=variant)
    try:
        x = synth.synth_frame(268, seed=seed).unsqueeze(0).to(dev)
        y = big.encode_latent(x, type='float')[0]
        e_y = rmse(sub(y, 499), gi["y_sub"])
        assert e_y <= 1e-5
        s = big._latent_side_frame(y[0])
        z_all = s["z_sym"].cpu().numpy().reshape(-1)
        med_np = big.entropy_bottleneck.quantiles.detach()[:, 0, 1].cpu().numpy().astype(np.float64)
        z_val = s["z"].double().cpu().numpy().reshape(256, -1) - med_np[:, None]
        n_zflip = _explained_flips(f"{which} z symbols", z_all, gi["z_sym_full"], z_val, lambda k: k + 0.5, 1e-4)
        sc_i, mu_i = _inject_reference_zhat(big, gi["z_sym_full"], dev)
        e_mi, e_si = rmse(sub(mu_i, 499), gi["means_sub"]), rmse(sub(sc_i, 499), gi["scales_sub"])
        assert e_mi <= 1e-5 and e_si <= 1e-5
        gcm = big.gaussian_conditional
        gci = ops.gaussian_conditional(sc_i, mu_i, gcm.scale_table, y=y[0].contiguous(), want=("idx", "sym"),
                                       scale_bound=big._scale_bound(), lik_bound=gcm.likelihood_bound)
        table = gcm.scale_table.double().cpu().numpy()
        sc_np = np.maximum(sc_i.double().cpu().numpy(), big._scale_bound())
        n_iflip = _explained_flips(f"{which} CDF indexes", gci["idx"].cpu().numpy(), gi["idx_full"], sc_np,
                                   lambda k: table[np.clip(k, 0, table.size - 1)], 1e-5)
        resid = (y[0].double() - mu_i.double()).cpu().numpy()
        n_sflip = _explained_flips(f"{which} y symbols", gci["sym"].cpu().numpy(), gi["sym_full"], resid, lambda k: k + 0.5, 1e-4)
        n_lat = gi["idx_full"].size
        assert n_iflip <= 16 and n_sflip <= max(8, 3 * n_lat * 2 * 0.8 * e_y)
        sr, raw, esc = ops.rans_resolve_symbols(torch.from_numpy(gi["sym_full"].astype(np.int32)).to(dev),
                                                torch.from_numpy(gi["idx_full"].astype(np.int32)).to(dev),
                                                gcm._quantized_cdf, gcm._cdf_length, gcm._offset)
        y_ref_stream = ops.rans_encode_resolved(sr.cpu().numpy(), raw.cpu().numpy(), esc.cpu().numpy())
        assert len(y_ref_stream) == int(gi["y_string_len"][0])
        assert hashlib.sha256(y_ref_stream).digest() == gi["y_string_sha256"].tobytes()
        # the compact records of the frame path on the same integers: same stream
        sr_c, rec_c, ovf = ops.rans_resolve_symbols_compact(torch.from_numpy(gi["sym_full"].astype(np.int32)).to(dev),
                                                            torch.from_numpy(gi["idx_full"].astype(np.int32)).to(dev),
                                                            gcm._quantized_cdf, gcm._cdf_length, gcm._offset)
        if float(ovf.float().sum()) == 0.0:
            assert ops.rans_encode_resolved_compact(sr_c.cpu().numpy(), rec_c.cpu().numpy()) == y_ref_stream
        # the product's own round trip of this frame: sizes in the reference's regime, lossless on its own integers
        out = big.compress(x)
        n_esc = big.last_n_escape()[0]
        assert abs(len(out["strings"][0][0]) - int(gi["y_string_len"][0])) <= 256
        assert abs(n_esc - int(gi["n_escape"][0])) <= 64
        if variant == "matched":
            assert n_esc < 0.01 * n_lat and len(out["strings"][0][0]) < 1.5e6
        if n_zflip == 0:
            assert out["strings"][1][0] == gi["z_string"].tobytes()
        y_hat = big.decompress(out["strings"], out["z_shape"], return_format='latent')
        assert torch.equal(torch.round(y_hat[0] - s["means"].reshape(y_hat[0].shape)).int().reshape(-1),
                           s["y_sym"].reshape(-1).int())
        print(f"{which}: y rmse {e_y:.2e}, z flips {n_zflip}, mu / sigma on the reference's z_hat {e_mi:.1e} / {e_si:.1e}, "
              f"index flips {n_iflip}, symbol flips {n_sflip} of {n_lat}; y stream {len(y_ref_stream)} bytes, "
              f"{int(gi['n_escape'][0])} escapes (product end to end: {len(out['strings'][0][0])} bytes, {n_esc} escapes)")
        ledger.ran(f"268 {which} (frame seed {seed}, {variant} weights): mu / sigma <= 1e-5 on the reference's z_hat, every "
                   "integer explained, sha256(y stream from the reference's integers) == reference-python-written stream's",
                   f"{len(y_ref_stream)} bytes, {int(gi['n_escape'][0])} escapes, {n_iflip} index / {n_sflip} symbol flips")
    finally:
        synth.apply_variant(big, seed=7, variant="default")


@pytest.fixture(scope="module")
def thin_m(dev, golden_dir):
    net = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(net, seed=7, variant="matched")
    net = net.to(dev)
    g = np.load(f"{golden_dir}/thin_e2e_m.npz")
    x = synth.synth_frame(8, seed=int(g["x_seed"][0])).unsqueeze(0).to(dev)
    y = net.encode_latent(x, type='float')[0]
    s = net._latent_side_frame(y[0])
    torch.cuda.synchronize()
    return net, g, x, y, s


def test_thin_matched_variant_vs_reference(thin_m, ledger):
    """The thin model under the entropy-matched weight variant (3.1 bits per latent, 7 escapes of 165 888 - a trained
    model's regime) against the reference run on the same weights / frame: integers, both streams, and the reference's
    own stream through the product's decompress()."""
    net, g, x, y, s = thin_m
    assert rmse(sub(y, 37), g["y_sub"]) <= 1e-5
    assert rmse(sub(s["means"], 37), g["means_sub"]) <= 1e-5 and rmse(sub(s["scales"], 37), g["scales_sub"]) <= 1e-5
    gc = net.gaussian_conditional
    assert np.array_equal(s["z_sym"].cpu().numpy().reshape(-1), g["z_sym_full"].astype(np.int32))
    table = gc.scale_table.double().cpu().numpy()
    sc_np = np.maximum(s["scales"].double().cpu().numpy(), net._scale_bound())
    n_iflip = _explained_flips("thin matched CDF indexes", s["idx"].cpu().numpy(), g["idx_full"], sc_np,
                               lambda k: table[np.clip(k, 0, table.size - 1)], 1e-5)
    resid = (y[0].double() - s["means"].double()).cpu().numpy()
    n_sflip = _explained_flips("thin matched y symbols", s["y_sym"].cpu().numpy(), g["sym_full"], resid, lambda k: k + 0.5, 1e-4)
    print(f"thin matched: index flips {n_iflip}, symbol flips {n_sflip} (reference margins y {g['margin_y'][0]:.1e}, "
          f"scale {g['margin_scale'][0]:.1e})")
    assert n_iflip <= 1 and n_sflip <= 1
    out = net.compress(x)
    assert out["strings"][1][0] == g["z_string"].tobytes()
    assert net.last_n_escape()[0] <= int(g["n_escape"][0]) + 2 < 100
    what = "thin matched variant (seed 162): y and z streams == reference-python-written streams; its .bin decodes to its x_hat"
    if n_iflip == 0 and n_sflip == 0:
        assert out["strings"][0][0] == g["y_string"].tobytes()
        strings = [[g["y_string"].tobytes()], [g["z_string"].tobytes()]]
        x_hat = net.decompress(strings, (18, 36))["x_hat"]
        e = rmse(sub(x_hat, 1009), g["xhat_rt_sub"])
        assert e <= 1e-5
        ledger.ran(what, f"{len(out['strings'][0][0])} + {len(out['strings'][1][0])} bytes, x_hat rmse {e:.1e}")
    else:
        ledger.not_applicable(what, f"{n_iflip} index / {n_sflip} symbol flips (each within tolerance of its boundary)")


def test_quality_159_vs_reference_golden(dev, golden_dir, ledger):
    """configs[1] of BASELINE.json: the 159-variable variant, encode_to_latent + latent_to_reconstruction,
    against the reference's own `VAEformer(0, ddconfig=... in_chans=159 ...)` on the same synthetic
    weights (tests/golden/make_golden.py --stage full159)."""
    g = np.load(f"{golden_dir}/full159.npz")
    net = vaeformer_pretrained(quality=159, pretrained=False)
    synth.load_synthetic(net, seed=3)
    net = net.to(dev)
    x = synth.synth_frame(159, seed=1).unsqueeze(0).to(dev)
    y = net.encode_latent(x, type='float')[0]
    assert y.shape == (1, 256, 72, 144) and torch.isfinite(y).all()
    e_y = rmse(sub(y, 499), g["y_sub"])
    assert abs(float(y.double().sum()) - g["y_stats"][0]) <= 1e-5 * y.numel()
    x_hat = net.decode_latent(synth_yhat(256, 5).to(dev))
    assert x_hat.shape == (1, 159, 721, 1440) and torch.isfinite(x_hat).all()
    e_x = rmse(sub(x_hat, 99991), g["xhat_sub"])
    print(f"159: y rmse {e_y:.3e}, x_hat rmse {e_x:.3e} (given identical y_hat)")
    assert e_y <= 1e-5 and e_x <= 1e-5
    assert rmse(x_hat[0, 0, 10].cpu(), g["xhat_row10_c0"]) <= 1e-5
    assert rmse(x_hat[0, 158, 720].cpu(), g["xhat_row720_c158"]) <= 1e-5
    # quantised latent: rounding flips are counted, not hidden.  y_hat = round(y - mu) + mu with mu from h_s(z_hat):
    # ONE flipped z symbol moves every mu by ~1e-3 through the global attention of h_s, so the gate is that of the
    # 268 test - with identical z symbols (the fixture holds all of them) mu / sigma / y_hat are held to 1e-5,
    # otherwise the bounded-flip branch; either way the counts are printed and land in the ledger.
    s = net._latent_side_frame(y[0])
    z_mis = int((s["z_sym"].cpu().reshape(-1).numpy() != g["z_sym"].astype(np.int32)).sum())
    z_hist = np.bincount((s["z_sym"].cpu().numpy().reshape(-1) + 64).clip(0, 128), minlength=129)
    assert int(np.abs(z_hist - g["z_sym_hist"]).sum()) <= 2 * z_mis
    idx_hist = np.bincount(s["idx"].cpu().numpy().reshape(-1), minlength=64)
    idx_l1 = int(np.abs(idx_hist - g["idx_hist"]).sum())
    idx_mis = int((sub(s["idx"], 499) != torch.from_numpy(g["idx_sub"])).sum())
    sym_mis = int((sub(s["y_sym"], 499) != torch.from_numpy(g["sym_sub"])).sum())
    e_z = rmse(sub(s["z"], 13), g["z_sub"])
    e_m, e_s = rmse(sub(s["means"], 499), g["means_sub"]), rmse(sub(s["scales"], 499), g["scales_sub"])
    y_hat = net.encode_latent(x, type='quantized')[1]
    d = (sub(y_hat, 499) - torch.from_numpy(g["y_hat_sub"])).abs()
    flips = int((d > 0.5).sum())
    e_q = float(torch.sqrt((d[d <= 0.5].double() ** 2).mean()))
    print(f"159: z rmse {e_z:.3e}, z symbol flips {z_mis}/{g['z_sym'].size} (reference margin {g['margin_z'][0]:.1e}), means "
          f"{e_m:.3e}, scales {e_s:.3e}, idx hist L1 {idx_l1}, sampled idx flips {idx_mis}, sampled symbol flips {sym_mis}, "
          f"y_hat symbol flips {flips}/{d.numel()}, y_hat rmse without them {e_q:.3e} (max {float(d[d <= 0.5].max()):.3e})")
    z_rms = float(np.sqrt(g["z_stats"][1] / s["z"].numel()))
    assert e_z <= 1e-5 * max(1.0, z_rms)
    assert z_mis <= max(4, int(3 * s["z"].numel() * 2 * 0.8 * e_z))      # flip probability 2|err| per element
    # every z flip is one step across a .5 boundary within 1e-4 (a flip further away is a bug, not float noise)
    med_np = net.entropy_bottleneck.quantiles.detach()[:, 0, 1].cpu().numpy().astype(np.float64)
    z_val = s["z"].double().cpu().numpy().reshape(med_np.size, -1) - med_np[:, None]
    assert _explained_flips("159 z symbols", s["z_sym"].cpu().numpy(), g["z_sym"], z_val, lambda k: k + 0.5, 1e-4) == z_mis
    if z_mis == 0:
        assert e_m <= 1e-5 and e_s <= 1e-5
        assert idx_mis <= 1 and sym_mis <= 1 and flips <= 1 and e_q <= 1e-5
    # (with z flips the end-to-end mu / sigma / y_hat of THIS run are not comparable - one flipped z symbol moves every mu
    # by ~1e-3 through h_s's global attention; the float comparison proper runs next, on the reference's z_hat, at 1e-5)
    # the reference's z_hat (all 165 888 symbols are in the fixture) injected into the product's h_s: means / scales /
    # y_hat held to the float tolerance on exactly the reference's input, whatever z flips the end-to-end run has
    sc_i, mu_i = _inject_reference_zhat(net, g["z_sym"], dev)
    e_mi, e_si = rmse(sub(mu_i, 499), g["means_sub"]), rmse(sub(sc_i, 499), g["scales_sub"])
    gci = ops.gaussian_conditional(sc_i, mu_i, net.gaussian_conditional.scale_table, y=y[0].contiguous(),
                                   want=("idx", "sym", "y_hat"), scale_bound=net._scale_bound(),
                                   lik_bound=net.gaussian_conditional.likelihood_bound)
    idx_i = int((sub(gci["idx"], 499) != torch.from_numpy(g["idx_sub"])).sum())
    sym_i = int((sub(gci["sym"], 499) != torch.from_numpy(g["sym_sub"])).sum())
    di = (sub(gci["y_hat"], 499) - torch.from_numpy(g["y_hat_sub"])).abs()
    e_qi = float(torch.sqrt((di[di <= 0.5].double() ** 2).mean()))
    ih = np.bincount(gci["idx"].cpu().numpy().reshape(-1), minlength=64)
    ih_l1 = int(np.abs(ih - g["idx_hist"]).sum())
    sh = np.bincount((gci["sym"].cpu().numpy().reshape(-1) + 256).clip(0, 512), minlength=513)
    sh_l1 = int(np.abs(sh - g["sym_hist"]).sum())
    print(f"159, reference z_hat injected: means {e_mi:.3e}, scales {e_si:.3e}, y_hat {e_qi:.3e}, sampled idx / symbol "
          f"flips {idx_i} / {sym_i}, index histogram L1 {ih_l1}, symbol histogram L1 {sh_l1}")
    assert e_mi <= 1e-5 and e_si <= 1e-5 and e_qi <= 1e-5
    assert idx_i <= 1 and sym_i <= 1 and int((di > 0.5).sum()) <= 1
    assert ih_l1 <= 32 and sh_l1 <= max(16, 2 * 3 * y.numel() * 2 * 0.8 * e_y)
    ledger.ran("159: means / scales / y_hat <= 1e-5 on the reference's z_hat (injected into the product's h_s)",
               f"means {e_mi:.1e}, y_hat {e_qi:.1e}; end-to-end z flips {z_mis}")


def test_api_268_channels_real_stats(big, dev, golden_dir, tmp_path):
    """cra5_api on the 268 model with the reference's real 268-long mean / std vectors
    (era5_stats_ref.npz = cra5_api.get_mean_std run by the reference's own code):
    encode_era5_as_bin(data=) -> decode_from_bin('de_normalized'), cra5_api.py:81-125,153-192."""
    from cra5_amd.api import cra5_api
    g = np.load(f"{golden_dir}/era5_stats_ref.npz")
    api = cra5_api(local_root=str(tmp_path), device="cuda", weights=big)
    assert np.array_equal(api.mean.cpu().numpy().reshape(-1), g["mean"])
    assert np.array_equal(api.std.cpu().numpy().reshape(-1), g["std"])
    assert [api.channels_to_vname[i] for i in range(268)] == list(g["vnames"])
    mean, std = torch.from_numpy(g["mean"]).view(268, 1, 1), torch.from_numpy(g["std"]).view(268, 1, 1)
    xn = synth.synth_frame(268, seed=2)
    frame = xn * std + mean                    # physical units
    ts = "2024-06-01T00:00:00"
    r = api.encode_era5_as_bin(ts, save_root=str(tmp_path / "CRA5"), data=frame)
    assert r["save_path"].endswith("CRA5/2024/2024-06-01T00:00:00.bin")
    # fused normalisation == the reference's explicit (x - mean) / std followed by compress()
    y_api = api.encode_to_latent(ts, data=frame)
    y_ref = big.encode_latent(((frame.to(dev) - api.mean) / api.std).unsqueeze(0), type='float')[0]
    assert rmse(y_api, y_ref) <= 1e-5
    e_gold = rmse(sub(y_api, 499), np.load(f"{golden_dir}/full268.npz")["y_sub"])
    print(f"268 api: y rmse vs reference golden (through de-normalise -> fused normalise) {e_gold:.3e}")
    assert e_gold <= 2e-5   # the physical-units round trip itself costs ~1 ulp of x
    d = api.decode_from_bin(ts, return_format='de_normalized')
    dn = api.decode_from_bin(ts, return_format='normalized')
    assert d["x_hat"].shape == (268, 721, 1440) or d["x_hat"].shape == (1, 268, 721, 1440)
    ref = dn["x_hat"].reshape(268, 721, 1440) * api.std + api.mean
    rel = ((d["x_hat"].reshape(268, 721, 1440) - ref).abs() / api.std).max()
    assert float(rel) <= 1e-4, float(rel)
    # per-variable error table in physical units (Readme.md:304-380 layout): finite, sane
    err = torch.sqrt(((d["x_hat"].reshape(268, 721, 1440) - frame.to(dev)) ** 2).mean(dim=(1, 2))).cpu()
    assert torch.isfinite(err).all()


def test_full268_reduced_precision_mode(big, dev, golden_dir):
    """BASELINE.json configs[4]: g_a / g_s on plain f16 operands (1 MFMA per product), entropy side
    fp32-accurate.  RMSE-tolerance gated against the fp32 reference golden (tolerances are the
    config's "~1e-2..1e-3" band; measured values are printed), and the stream must still decode
    to exactly the y_hat the encoder quantised."""
    g = np.load(f"{golden_dir}/full268.npz")
    x = synth.synth_frame(268, seed=2).unsqueeze(0).to(dev)
    big.precision = "f16"
    try:
        y = big.encode_latent(x, type='float')[0]
        e_y = rmse(sub(y, 499), g["y_sub"])
        y_rms = float(torch.sqrt((torch.from_numpy(g["y_sub"]).double() ** 2).mean()))
        x_hat = big.decode_latent(synth_yhat(256, 5).to(dev))
        e_x = rmse(sub(x_hat, 99991), g["xhat_sub"])
        x_rms = float(torch.sqrt((torch.from_numpy(g["xhat_sub"]).double() ** 2).mean()))
        print(f"268 f16 mode: y rmse {e_y:.3e} (rms {y_rms:.3f}), x_hat rmse {e_x:.3e} (rms {x_rms:.3f})")
        assert e_y <= 1e-2 * max(1.0, y_rms)
        assert e_x <= 1e-2 * max(1.0, x_rms)
        assert e_y > 1e-5            # it really is the reduced-precision path
        out = big.compress_from_latent(y)
        s = big._latent_side_frame(y[0], want_lik=False)
        y_hat = big.decompress(out["strings"], out["z_shape"], return_format='latent')
        assert torch.equal(y_hat[0].reshape(-1), s["y_hat"].reshape(-1))
    finally:
        big.precision = "fp32"
    # and the default mode is untouched afterwards
    y2 = big.encode_latent(x, type='float')[0]
    assert rmse(sub(y2, 499), g["y_sub"]) <= 1e-5


def test_reduced_precision_plain_layout_equals_the_split_layout(big, thin, dev):
    """Round 5: in the reduced-precision mode activations / weights of g_a and g_s travel as PLAIN f16 rows wherever the
    consuming kernel takes them (VAEformer.f16_layout = "plain", the default) - a different memory layout of the same f16 values:
    y, x_hat and the streams equal those of the split layout (rounds 1-4) BIT FOR BIT, on the 268 model (every GEMM of
    a block plain) and on the thin one (only its fc1 / fc2 are wide enough: the mixed case)."""
    for net, C, L in ((big, 268, 256), (thin, 8, 16)):
        x = synth.synth_frame(C, seed=6).unsqueeze(0).to(dev)
        keep = (net.precision, net.f16_layout)
        res = {}
        try:
            net.precision = "f16"
            for lay in ("split", "plain"):
                net.f16_layout = lay
                y = net.encode_latent(x, type='float')[0]
                x_hat = net.decode_latent(synth_yhat(L, 5).to(dev))
                out = net.compress(x)
                res[lay] = (y.clone(), x_hat.clone(), out["strings"])
        finally:
            net.precision, net.f16_layout = keep
        assert torch.equal(res["plain"][0], res["split"][0]), C
        assert torch.equal(res["plain"][1], res["split"][1]), C
        assert res["plain"][2] == res["split"][2], C
        assert bool(torch.isfinite(res["plain"][1]).all())


def test_thin_batch_of_two_frames(thin, dev):
    """B > 1 through the public API: strings come back per frame (vaeformer.py:399-401 layout
    [y_strings, z_strings], each a list of B byte strings) and every frame equals its B=1 result."""
    xs = torch.stack([synth.synth_frame(thin.cfg['in_chans'], seed=s) for s in (2, 9)]).to(dev)
    out = thin.compress(xs)
    assert len(out["strings"]) == 2 and len(out["strings"][0]) == 2 and len(out["strings"][1]) == 2
    for b in range(2):
        one = thin.compress(xs[b:b + 1])
        assert one["strings"][0][0] == out["strings"][0][b]
        assert one["strings"][1][0] == out["strings"][1][b]
    rec = thin.decompress(out["strings"], out["z_shape"])["x_hat"]
    assert rec.shape == xs.shape
    for b in range(2):
        one = thin.decompress([[out["strings"][0][b]], [out["strings"][1][b]]], out["z_shape"])["x_hat"]
        assert torch.equal(one[0], rec[b])


def test_frame_pipeline_scheduling_is_transparent(thin, dev):
    """Frames in flight, GPU-phase slots and exclusive phases only change WHEN kernels run: every
    frame's byte streams and reconstruction equal the serial single-frame result, in order."""
    from cra5_amd.pipeline import FramePipeline
    frames = [synth.synth_frame(thin.cfg['in_chans'], seed=s).unsqueeze(0).to(dev) for s in (2, 9, 4)]
    ref = []
    for f in frames:
        out = thin.compress(f)
        ref.append((out["strings"], thin.decompress(out["strings"], out["z_shape"])["x_hat"]))
    keep = (thin.gpu_exclusive, thin.gpu_slots)
    pipe = FramePipeline(thin, workers=4)
    try:
        for excl, slots in ((False, 0), (False, 2), (True, 0)):
            thin.gpu_exclusive, thin.gpu_slots = excl, slots
            res = pipe.roundtrip([frames[i % 3] for i in range(7)])
            for i, (out, x_hat) in enumerate(res):
                strings, x_ref = ref[i % 3]
                assert out["strings"] == strings
                assert torch.equal(x_hat, x_ref)
    finally:
        thin.gpu_exclusive, thin.gpu_slots = keep
        pipe.close()


def test_gpu_resolved_encoder_same_stream(thin, big, dev):
    """Resolving the y symbols against the CDF tables on the device (cra5_rans_resolve_symbols_i32 +
    cra5_rans_encode_resolved) writes exactly the stream of the table-driven host encoder, on the thin
    model and on the full-size one (escape-heavy: half of its symbols sit in the narrowest row)."""
    for net, C in ((thin, thin.cfg['in_chans']), (big, 268)):
        x = synth.synth_frame(C, seed=11).unsqueeze(0).to(dev)
        keep = net.resolve_on_gpu
        try:
            net.resolve_on_gpu = True
            a = net.compress(x)
            net.resolve_on_gpu = False
            b = net.compress(x)
        finally:
            net.resolve_on_gpu = keep
        assert a["strings"][0][0] == b["strings"][0][0] and a["strings"][1][0] == b["strings"][1][0]
        assert len(a["strings"][0][0]) > 1000


def test_rate_estimate_predicts_stream_size(thin, dev):
    """forward()'s likelihoods (vaeformer.py:302-333) -> estimated bits (rate_distortion.py:71-74) vs the
    bytes compress() really writes.  With trained weights the range coder lands within ~1 % of the model
    entropy; with the synthetic weights used here many residuals fall outside the CDF tables, where the
    likelihood floor (1e-9 = 30 bits) over-estimates what the escape code spends (bin + a few nibbles), so
    the streams come out SHORTER than the estimate (measured -18 %): gate loosely, both ways."""
    from cra5_amd import metrics
    x = synth.synth_frame(thin.cfg['in_chans'], seed=6).unsqueeze(0).to(dev)
    out = thin.forward(x)
    bits = metrics.estimated_bits(out)
    s = thin.compress(x)["strings"]
    real = 8.0 * (len(s[0][0]) + len(s[1][0]))
    print(f"thin: estimated {bits / 8:.0f} bytes, coded {real / 8:.0f} bytes ({100 * (real / bits - 1):+.1f} %)")
    assert 0.6 * bits <= real <= 1.25 * bits
    assert metrics.estimated_bpp(out, 721 * 1440) == pytest.approx(bits / (721 * 1440))


def test_api_batch_methods_pinned_pipeline(thin, dev, tmp_path):
    """encode_era5_batch / decode_batch (frames streamed through the frame pipeline with pinned host staging):
    every frame's .bin and reconstruction equal the single-frame cra5_api methods'."""
    from cra5_amd.api import cra5_api
    api = cra5_api(local_root=str(tmp_path), device="cuda", weights=thin)
    api._mean_flat = torch.linspace(-1, 1, 8, device=dev)
    api._std_flat = torch.linspace(0.5, 2, 8, device=dev)
    api.mean, api.std = api._mean_flat.view(8, 1, 1), api._std_flat.view(8, 1, 1)
    frames = [(synth.synth_frame(8, seed=s) * api.std.cpu() + api.mean.cpu()).numpy() for s in (3, 4, 5, 6, 7)]
    stamps = [f"2024-06-01T{h:02d}:00:00" for h in range(5)]
    res = api.encode_era5_batch(stamps, data=frames, save_root=str(tmp_path / "CRA5"), workers=3)
    assert [r["save_path"].rsplit("/", 1)[1] for r in res] == [s + ".bin" for s in stamps]
    out = np.empty((5, 8, 721, 1440), dtype=np.float32)
    rec = api.decode_batch(stamps, out=out, workers=3)
    rec_n = api.decode_batch(stamps[:2], return_format="normalized", workers=2)
    for i, ts in enumerate(stamps):
        one = api.encode_era5_as_bin(ts, save_root=str(tmp_path / "single"), data=frames[i])
        assert open(one["save_path"], "rb").read() == open(res[i]["save_path"], "rb").read()
        d = api.decode_from_bin(ts, return_format="de_normalized")["x_hat"]
        assert np.array_equal(d.cpu().numpy().reshape(8, 721, 1440), out[i]) and rec[i] is not None
    dn = api.decode_from_bin(stamps[1], return_format="normalized")["x_hat"]
    assert np.array_equal(dn.cpu().numpy().reshape(8, 721, 1440), rec_n[1])
    # roundtrip_batch: the test.py loop as a stream - same files, same reconstructions
    rt = api.roundtrip_batch(stamps, data=frames, save_root=str(tmp_path / "RT"), workers=3)
    for i in range(5):
        assert open(rt[i][0]["save_path"], "rb").read() == open(res[i]["save_path"], "rb").read()
        assert np.array_equal(rt[i][1], out[i])
    # sink=: the consumer sees each frame in the decoding thread's pinned buffer (no [n, C, H, W] host array)
    sums = api.decode_batch(stamps, workers=3, sink=lambda i, a: (i, float(a.astype(np.float64).sum())))
    assert sums == [(i, float(out[i].astype(np.float64).sum())) for i in range(5)]
    # host staging takes what the reference's xarray path can hand over: float64 arrays, CPU tensors, non-contiguous
    # views - same bytes as the float32 frame
    f64 = frames[2].astype(np.float64)
    strided = np.ascontiguousarray(frames[2].transpose(0, 2, 1)).transpose(0, 2, 1)
    alt = api.encode_era5_batch(["2024-06-02T00:00:00", "2024-06-02T01:00:00", "2024-06-02T02:00:00"],
                                data=[f64, torch.from_numpy(frames[2]), strided], save_root=str(tmp_path / "alt"), workers=2)
    want = open(res[2]["save_path"], "rb").read()
    assert all(open(r["save_path"], "rb").read() == want for r in alt)
    # the link gate (round 6: one frame per direction on the host link at a time) is scheduling, not arithmetic: with it
    # off, and with out= copy teams, the same files and reconstructions come back; bad `out=` arrays are refused
    assert api.link_serial is True and api.runtime.link_serial is True
    api.link_serial = False
    try:
        rt2 = api.roundtrip_batch(stamps, data=frames, save_root=str(tmp_path / "RT2"), workers=3)
    finally:
        api.link_serial = True
    for i in range(5):
        assert open(rt2[i][0]["save_path"], "rb").read() == open(res[i]["save_path"], "rb").read()
        assert np.array_equal(rt2[i][1], out[i])
    for bad in (np.empty((5, 8, 721, 1440), dtype=np.float64), np.empty((4, 8, 721, 1440), dtype=np.float32),
                np.empty((5, 8, 1440, 721), dtype=np.float32).transpose(0, 1, 3, 2)):
        with pytest.raises(ValueError):
            api.decode_batch(stamps, out=bad, workers=2)
        with pytest.raises(ValueError):
            api.roundtrip_batch(stamps, data=frames, save_root=str(tmp_path / "RT3"), workers=2, out=bad)


def test_hyper_prior_engine_is_pinned_across_engine_settings(thin, thin_side, dev):
    """the decoder re-derives every CDF index from h_s(z_hat); the engine behind h_a / h_s must not
    depend on CRA5_GEMM / CRA5_ATTN / the precision mode, or a stream written under one setting desynchronises
    when read under another.  Encode with the default engines, decode with the exact-f32 engines and with the
    reduced-precision mode selected: the latent that comes back is bit-identical every time."""
    x, y, s = thin_side
    out = thin.compress_from_latent(y)
    keep = (thin.gemm_mode, thin.attn_mode, thin.precision)
    try:
        for gm, am, pr in (("f32", "f32", "fp32"), ("split", "f32", "fp32"), ("split", "split", "f16")):
            thin.gemm_mode, thin.attn_mode, thin.precision = gm, am, pr
            y_hat = thin.decompress(out["strings"], out["z_shape"], return_format='latent')
            assert torch.equal(y_hat[0].reshape(-1), s["y_hat"].reshape(-1)), (gm, am, pr)
            s2 = thin._latent_side_frame(y[0])
            assert torch.equal(s2["idx"], s["idx"]) and torch.equal(s2["y_sym"], s["y_sym"]), (gm, am, pr)
    finally:
        thin.gemm_mode, thin.attn_mode, thin.precision = keep


def test_full268_pipeline_distinct_frames_equal_serial(big, dev):
    """BASELINE configs[3] in miniature on one GPU: distinct frames x_f ~ N(0,1) (seed 1000 + f) through the frame
    pipeline, several in flight on their own streams - every frame's two byte streams equal its serial compress()."""
    from cra5_amd.pipeline import FramePipeline
    g = torch.Generator(device=dev)
    frames = []
    for f in range(3):
        g.manual_seed(1000 + f)
        frames.append(torch.randn((1, 268, 721, 1440), generator=g, device=dev))
    serial = [big.compress(f)["strings"] for f in frames]
    assert len({s[0][0] for s in serial}) == 3                     # the frames really differ
    keep = (big.gpu_exclusive, big.gpu_slots)
    pipe = FramePipeline(big, workers=3)
    try:
        big.gpu_exclusive, big.gpu_slots = False, 2
        outs = pipe.compress([frames[i % 3] for i in range(6)])
        for i, o in enumerate(outs):
            assert o["strings"] == serial[i % 3]
        rec = pipe.decompress(outs[:3])
        for i in range(3):
            ref = big.decompress(serial[i], outs[i]["z_shape"])["x_hat"]
            assert torch.equal(rec[i]["x_hat"], ref)
    finally:
        big.gpu_exclusive, big.gpu_slots = keep
        pipe.close()


# --------------------------------------------------------------------------------------
# range guard: a checkpoint whose activations leave the f16 range must never give a quietly
# clipped frame - csrc/split.h poisons the out-of-range element, the model re-runs the frame on the exact-f32 engines
# --------------------------------------------------------------------------------------


def _stress_thin(dev, key_mod):
    """Thin model with synthetic weights + an outlier-channel modification (`key_mod(net)` runs on the host copy)."""
    net = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(net, seed=7)
    with torch.no_grad():
        key_mod(net)
    return net.to(dev)


def test_range_guard_reruns_an_overflowing_encode_on_the_exact_f32_engines(dev):
    """Massive-activation hidden units in an encoder MLP (four fc1 rows x 3e5: their GELU outputs are ~1e5..1e6, beyond
    f16's 65 504): the split GELU epilogue poisons them, y comes out non-finite, the guard re-runs the frame on the
    exact-f32 engines.  The result is the pure exact-f32 run's, bit for bit, and matches the CPU oracle."""
    def mod(net):       # fc2 takes the factor back out: the block's output stays O(1), only the hidden units are huge
        net.g_a.blocks[3].mlp.fc1.weight[:4] *= 3e5
        net.g_a.blocks[3].mlp.fc1.bias[:4] *= 3e5
        net.g_a.blocks[3].mlp.fc2.weight[:, :4] /= 3e5
    net = _stress_thin(dev, mod)
    x = synth.synth_frame(8, seed=2).unsqueeze(0).to(dev)
    with pytest.warns(RuntimeWarning, match="exact-f32"):
        out = net.compress(x)
    assert net.range_fallbacks == [1, 0]
    rec = net.decompress(out["strings"], out["z_shape"])["x_hat"]      # g_s is untouched: split engines, no fallback
    assert torch.isfinite(rec).all() and net.range_fallbacks == [1, 0]
    with pytest.warns(RuntimeWarning):
        y = net.encode_latent(x, type='float')[0]
    assert torch.isfinite(y).all()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    y_ref = R.encode_y(x.cpu(), sd, R.cfg_thin())
    e = rmse(y, y_ref) / max(1.0, float(y_ref.double().pow(2).mean().sqrt()))
    print(f"range guard: y (exact-f32 re-run) vs CPU oracle: relative rmse {e:.2e}")
    assert e <= 1e-5
    # the pure exact-f32 engines give the same streams (the re-run IS that engine)
    ref_net = _stress_thin(dev, mod)
    ref_net.gemm_mode, ref_net.attn_mode = "f32", "f32"
    out_f32 = ref_net.compress(x)
    assert ref_net.range_fallbacks == [0, 0]
    assert out_f32["strings"] == out["strings"]
    # guard off: an error, never a quietly clipped frame
    net.range_guard = False
    with pytest.raises(FloatingPointError, match="f16 range"):
        net.compress(x)


def test_range_guard_decode_side_and_pinned_hyper_path(dev):
    """The same on the decode side (outlier rows in a g_s MLP: decompress() re-runs g_s on the exact-f32 engines and
    matches the oracle), and in the hyper-prior path, whose engine is pinned on both sides of the codec: an error."""
    base = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(base, seed=7)
    base = base.to(dev)
    x = synth.synth_frame(8, seed=2).unsqueeze(0).to(dev)
    out = base.compress(x)                                   # a valid stream pair of the unmodified entropy side

    def mod_gs(net):
        net.g_s.blocks[2].mlp.fc1.weight[:4] *= 3e5
        net.g_s.blocks[2].mlp.fc1.bias[:4] *= 3e5
        net.g_s.blocks[2].mlp.fc2.weight[:, :4] /= 3e5
    net = _stress_thin(dev, mod_gs)
    with pytest.warns(RuntimeWarning, match="exact-f32"):
        rec = net.decompress(out["strings"], out["z_shape"])["x_hat"]
    assert net.range_fallbacks == [0, 1] and torch.isfinite(rec).all()
    y_hat = net.decompress(out["strings"], out["z_shape"], return_format='latent')
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x_ref = R.g_s(y_hat.cpu(), sd, R.cfg_thin())
    e = rmse(rec, x_ref) / max(1.0, float(x_ref.double().pow(2).mean().sqrt()))
    print(f"range guard: x_hat (exact-f32 re-run) vs CPU oracle: relative rmse {e:.2e}")
    assert e <= 1e-5

    def mod_hs(net):
        net.h_s.blocks[1].mlp.fc1.weight[:4] *= 3e5
        net.h_s.blocks[1].mlp.fc1.bias[:4] *= 3e5
        net.h_s.blocks[1].mlp.fc2.weight[:, :4] /= 3e5
    bad = _stress_thin(dev, mod_hs)
    with pytest.raises(FloatingPointError, match="hyper-prior"):
        bad.compress(x)
    with pytest.raises(FloatingPointError, match="hyper-prior"):
        bad.decompress(out["strings"], out["z_shape"])
    # a non-finite input frame is not a range event: both engines see it, the error says so
    xn = x.clone()
    xn[0, 3, 100, 200] = float("nan")
    with pytest.warns(RuntimeWarning), pytest.raises(FloatingPointError, match="input frame"):
        base.compress(xn)


# --------------------------------------------------------------------------------------
# compact device <-> host records: same streams, same reconstructions, 32-bit fallback
# --------------------------------------------------------------------------------------


def test_compact_records_device_kernels_match_the_32_bit_ones(thin, thin_side, dev):
    """resolve_symbols_compact_kernel / gaussian_conditional_compact_kernel against the int32 kernels they shadow, on a
    real frame's integers."""
    _, y, s = thin_side
    gc = thin.gaussian_conditional
    sym, idx = s["y_sym"].reshape(-1).contiguous(), s["idx"].reshape(-1).contiguous()
    sr, raw, esc = ops.rans_resolve_symbols(sym, idx, gc._quantized_cdf, gc._cdf_length, gc._offset)
    sr2, rec, ovf = ops.rans_resolve_symbols_compact(sym, idx, gc._quantized_cdf, gc._cdf_length, gc._offset)
    torch.cuda.synchronize()
    assert int(ovf[0]) == 0 and torch.equal(sr, sr2)
    rec32 = rec.cpu().numpy().view(np.uint16).astype(np.uint32)
    assert np.array_equal(rec32 >> 12, esc.cpu().numpy().astype(np.uint32))          # 1 + nibble count, 0 for regular symbols
    assert np.array_equal(rec32 & 0xFFF, raw.cpu().numpy().view(np.uint32))
    a = ops.rans_encode_resolved(sr.cpu().numpy(), raw.cpu().numpy(), esc.cpu().numpy())
    assert ops.rans_encode_resolved_compact(sr2.cpu().numpy(), rec.cpu().numpy()) == a
    sc, mu = s["scales"].contiguous(), s["means"].contiguous()
    c = ops.gaussian_conditional_compact(sc, mu, gc.scale_table, want_idx8=True, scale_bound=thin._scale_bound())
    assert torch.equal(c["idx8"].int().reshape(-1), idx)
    yh = ops.gaussian_conditional_compact(None, mu, sym16_in=s["y_sym"].to(torch.int16).contiguous())["y_hat"]
    assert torch.equal(yh.reshape(-1), s["y_hat"].reshape(-1))


def test_compact_records_same_streams_and_reconstruction_incl_the_32_bit_fallback(thin, thin_side, dev):
    x, y, s = thin_side
    assert thin.compact_records
    out_c = thin.compress(x)
    rec_c = thin.decompress(out_c["strings"], out_c["z_shape"])["x_hat"]
    thin.compact_records = False
    try:
        out_w = thin.compress(x)
        rec_w = thin.decompress(out_c["strings"], out_c["z_shape"])["x_hat"]
    finally:
        thin.compact_records = True
    assert out_c["strings"] == out_w["strings"] and torch.equal(rec_c, rec_w)
    assert thin.last_n_escape()[0] > 0
    # a latent with symbols far outside every table row (escape payloads beyond 12 bits on the encode side, symbols beyond
    # int16 on the decode side): both sides fall back to the 32-bit records for that frame, the round trip stays exact
    y_big = y.clone()
    y_big[0, 3, 10, 20] += 40000.0          # (inside the f16 range of the h_a input split: no poison, only huge symbols)
    y_big[0, 5, 11, 21] -= 40000.0
    out_b = thin.compress_from_latent(y_big)
    y_hat = thin.decompress(out_b["strings"], out_b["z_shape"], return_format='latent')
    thin.compact_records = False
    try:
        out_b32 = thin.compress_from_latent(y_big)
        y_hat32 = thin.decompress(out_b["strings"], out_b["z_shape"], return_format='latent')
    finally:
        thin.compact_records = True
    assert out_b["strings"] == out_b32["strings"] and torch.equal(y_hat, y_hat32)
    assert abs(float(y_hat[0, 3, 10, 20] - y_big[0, 3, 10, 20])) <= 0.5 + 1e-3
    assert abs(float(y_hat[0, 5, 11, 21] - y_big[0, 5, 11, 21])) <= 0.5 + 1e-3
