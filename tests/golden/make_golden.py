"""Golden-vector generator (BUILD CONTAINER ONLY - needs /root/reference).

Imports the reference's own Python through oracle/ref_shim.py, loads the
deterministic synthetic weights of cra5_amd/synth.py into the reference model and
dumps small input/output fixtures (data only) next to this script:

  pmf_cdf.json          outputs of the reference's ops.cpp (compiled as oracle/_ref/_CXX)
  state_keys.json       state-dict key -> shape of the 268 model and the thin model
  tables_default.npz    GaussianConditional.update() tables for get_scale_table() and
                        EntropyBottleneck.update() tables at the synthetic EB params
  ops_small.npz         per-op outputs (window attention x3 incl. padding, global
                        attention hd=72, block, patch-embed, un-embed, hyper un-embed)
  thin_e2e.npz          full-spatial thin model: every stage of encode / latent side /
                        decode, sub-sampled, + rANS strings
  full268.npz           (--stage full; ~3 min, ~18 GB RAM) the real 268 architecture
  full159.npz           (--stage full159) BASELINE configs[1]: the same architecture with 159
                        variables (reference `VAEformer(0, ddconfig=... in_chans=159 ...)`,
                        the ddconfig route of vaeformer.py:78-143; `VAEformer(159)` itself
                        crashes in the reference), encode_latent + decode_latent
  cnn_zoo.npz           (--stage cnn) bmshj2018-factorized / -hyperprior, mbt2018-mean at N=32, M=48 on a
                        3 x 128 x 192 image: y, z, h_s, x_hat (forward, round trip, synthetic y_hat),
                        likelihood bits, rANS strings
  era5_stats_ref.npz    (--stage stats) the reference's own 268-long mean / std vectors
                        (cra5_api.get_mean_std, cra5_api.py:243-261) and channel -> vname map
                        (:228-241), computed by the reference's code on its own JSONs/config
  thin_fp64.npz / full268_fp64.npz  (--stage thin64 / full64) the reference run in float64:
                        calibration of the fp32 noise floor of the reference itself

Usage:  python tests/golden/make_golden.py --stage small thin [full]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

ref_shim.install()

from cra5.models.vaeformer.vaeformer import VAEformer  # noqa: E402  (reference)
from cra5.models.vaeformer import vit_nlc  # noqa: E402  (reference)
from cra5.models.compressai.entropy_models import GaussianConditional  # noqa: E402
from cra5.models.compressai.models.base import get_scale_table  # noqa: E402
from cra5.models.compressai.layers.gdn import GDN  # noqa: E402
import cra5.models.compressai._CXX as REF_CXX  # noqa: E402

from cra5_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)

THIN_DD_KW = dict(z_dim=None, learnable_pos=True, window=True, window_size=[(24, 24), (12, 48), (48, 12)],
                  interval=4, drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                  test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=True,
                  img_size=(721, 1440), embed_dim=128, depth=8, num_heads=2)
THIN_PRIOR_KW = dict(z_dim=16, embed_dim=144, depth=4, num_heads=2, interval=1, learnable_pos=True,
                     window=False, drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                     test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=False,
                     img_size=(72, 144))


def build_thin():
    return VAEformer(0, embed_dim=16, z_channels=16, y_channels=128, sample_posterior=False,
                     frozen_encoder=False, lower_dim=True,
                     ddconfig=dict(arch='vit_base', patch_size=(11, 10), patch_stride=(10, 10), in_chans=8,
                                   out_chans=8, pretrained_model='', kwargs=dict(THIN_DD_KW)),
                     priorconfig=dict(patch_size=(4, 4), in_chans=16, out_chans=16, pretrained_model='',
                                      kwargs=dict(THIN_PRIOR_KW))).eval()


def load_synth(net, seed, variant="default"):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed, variant=variant)
    missing, unexpected = torch.nn.Module.load_state_dict(net, sd, strict=False)
    assert not unexpected, unexpected
    net.update(force=True)
    return shapes


def sub(t, step):
    return t.detach().reshape(-1)[::step].clone().numpy()


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.abs().max().item()], dtype=np.float64)


def stage_small():
    # ---- pmf -> cdf ---------------------------------------------------------
    rng = np.random.default_rng(1234)
    cases = []
    for t in range(40):
        n = int(rng.integers(1, 64))
        p = rng.random(n).astype(np.float32) ** int(rng.integers(1, 9))
        p /= p.sum()
        if t % 4 == 0 and n > 3:
            p[rng.integers(0, n, size=n // 2)] = 0.0
        if t % 7 == 0:
            p = (p * 0.37).astype(np.float32)  # un-normalised
        if float(p.sum()) == 0.0:
            p[0] = 1.0
        cases.append(dict(pmf=[float(v) for v in p], precision=16,
                          cdf=[int(v) for v in REF_CXX.pmf_to_quantized_cdf([float(v) for v in p], 16)]))
    cases.append(dict(pmf=[1.0], precision=16, cdf=[int(v) for v in REF_CXX.pmf_to_quantized_cdf([1.0], 16)]))
    tiny = [1e-9] * 30 + [1.0]
    cases.append(dict(pmf=tiny, precision=16, cdf=[int(v) for v in REF_CXX.pmf_to_quantized_cdf(tiny, 16)]))
    errs = []
    for bad in ([-0.1, 0.5], [float("nan"), 1.0], [float("inf")], [0.0, 0.0]):
        try:
            REF_CXX.pmf_to_quantized_cdf(bad, 16)
            errs.append(dict(pmf=[repr(v) for v in bad], raises=False))
        except Exception as e:  # noqa: BLE001
            errs.append(dict(pmf=[repr(v) for v in bad], raises=True, type=type(e).__name__))
    json.dump(dict(cases=cases, errors=errs), open(os.path.join(HERE, "pmf_cdf.json"), "w"))

    # ---- state-dict layouts ---------------------------------------------------
    thin = build_thin()
    thin_shapes = {k: list(v.shape) for k, v in thin.state_dict().items()}
    big = VAEformer(268)
    big_shapes = {k: list(v.shape) for k, v in big.state_dict().items()}
    del big
    json.dump(dict(thin=thin_shapes, v268=big_shapes), open(os.path.join(HERE, "state_keys.json"), "w"))

    # ---- default tables -------------------------------------------------------
    gc = GaussianConditional(None)
    gc.update_scale_table(get_scale_table(), force=True)
    load_synth(thin, seed=7)
    eb = thin.entropy_bottleneck
    np.savez_compressed(os.path.join(HERE, "tables_default.npz"),
                        scale_table=gc.scale_table.numpy(), gc_cdf=gc._quantized_cdf.numpy(),
                        gc_len=gc._cdf_length.numpy(), gc_off=gc._offset.numpy(),
                        eb_cdf=eb._quantized_cdf.numpy(), eb_len=eb._cdf_length.numpy(),
                        eb_off=eb._offset.numpy())

    # ---- per-op fixtures ------------------------------------------------------
    out = {}
    vit_nlc.COMPAT = False
    H, W, C, heads = 72, 144, 128, 2
    g = torch.Generator().manual_seed(11)
    xtok = torch.randn(1, H * W, C, generator=g)
    out["tok_in_step"] = np.array([1])
    for name, ws in (("w24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
        m = vit_nlc.WindowAttention(C, ws, heads, qkv_bias=True).eval()
        shapes = {f"attn.{k}": tuple(v.shape) for k, v in m.state_dict().items()}
        sd = synth.fill_state_dict(shapes, seed=21)
        m.load_state_dict({k[5:]: v for k, v in sd.items()})
        out[f"winattn_{name}"] = sub(m(xtok, H, W), 61)
    m = vit_nlc.Attention(144, num_heads=2, qkv_bias=True).eval()
    sd = synth.fill_state_dict({f"attn.{k}": tuple(v.shape) for k, v in m.state_dict().items()}, seed=22)
    m.load_state_dict({k[5:]: v for k, v in sd.items()})
    x648 = torch.randn(1, 648, 144, generator=g)
    out["globattn_hd72"] = sub(m(x648, 18, 36), 7)
    m = vit_nlc.Attention(C, num_heads=heads, qkv_bias=True).eval()
    sd = synth.fill_state_dict({f"attn.{k}": tuple(v.shape) for k, v in m.state_dict().items()}, seed=23)
    m.load_state_dict({k[5:]: v for k, v in sd.items()})
    out["globattn_hd64"] = sub(m(xtok, H, W), 61)
    for name, ws, win in (("blk_w48x12", (48, 12), True), ("blk_glob", (72, 144), False)):
        m = vit_nlc.Block(C, heads, mlp_ratio=4, qkv_bias=True, norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6),
                          window_size=ws, window=win).eval()
        sd = synth.fill_state_dict({f"blocks.0.{k}": tuple(v.shape) for k, v in m.state_dict().items()}, seed=24)
        m.load_state_dict({k[9:]: v for k, v in sd.items()})
        out[name] = sub(m(xtok, H, W), 61)
    # GDN / IGDN
    for inv in (False, True):
        m = GDN(12, inverse=inv).eval()
        gg = torch.Generator().manual_seed(31)
        m.beta.data = m.beta.data + 0.3 * torch.rand(12, generator=gg)
        m.gamma.data = m.gamma.data + 0.05 * torch.rand(12, 12, generator=gg)
        xg = torch.randn(2, 12, 9, 7, generator=gg)
        out[f"gdn_inv{int(inv)}_beta"] = m.beta.data.numpy()
        out[f"gdn_inv{int(inv)}_gamma"] = m.gamma.data.numpy()
        out[f"gdn_inv{int(inv)}_x"] = xg.numpy()
        out[f"gdn_inv{int(inv)}_y"] = m(xg).numpy()
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **out)
    print("small done")


def run_e2e(net, x, yhat_synth, step_lat, step_img, tag, full_ints=False):
    """All stages of the reference path on `x`; returns dict of sub-sampled arrays.
    full_ints: also keep EVERY CDF index (int8) and y symbol (int16), so that a consumer can compare the
    integer side element by element instead of through histograms."""
    o = {}
    t0 = time.time()
    moments = net.quant_conv(net.g_a(x))
    y = moments[:, : moments.shape[1] // 2]
    print(tag, "g_a", time.time() - t0)
    o["y_sub"], o["y_stats"] = sub(y, step_lat), stats(y)
    z = net.h_a(y)
    o["z"] = z.numpy().astype(np.float32) if z.numel() < 20000 else sub(z, 13)
    o["z_stats"] = stats(z)
    z_hat, z_lik = net.entropy_bottleneck(z)
    o["z_lik_sub"] = sub(z_lik, 13)
    z_strings = net.entropy_bottleneck.compress(z)
    z_hat2 = net.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
    assert torch.equal(z_hat, z_hat2)
    o["z_sym"] = net.entropy_bottleneck.quantize(
        z, "symbols", net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)).numpy().astype(np.int32)
    if o["z_sym"].size > 20000:
        o["z_sym_hist"] = np.bincount((o["z_sym"].reshape(-1) + 64).clip(0, 128), minlength=129)
        if full_ints:       # EVERY z symbol: a consumer can inject the reference's z_hat into its own h_s
            assert np.abs(o["z_sym"]).max() < 32768
            o["z_sym_full"] = o["z_sym"].reshape(-1).astype(np.int16)
        o["z_sym"] = o["z_sym"].reshape(-1)[::13]
    params = net.h_s(z_hat)
    scales, means = params.chunk(2, 1)
    o["scales_sub"], o["means_sub"] = sub(scales, step_lat), sub(means, step_lat)
    o["scales_stats"], o["means_stats"] = stats(scales), stats(means)
    idx = net.gaussian_conditional.build_indexes(scales)
    o["idx_sub"] = sub(idx, step_lat).astype(np.int32)
    o["idx_hist"] = np.bincount(idx.reshape(-1).numpy(), minlength=64)
    y_hat, y_lik = net.gaussian_conditional(y, scales, means=means)
    o["y_lik_sub"] = sub(y_lik, step_lat)
    sym = net.gaussian_conditional.quantize(y, "symbols", means)
    o["sym_sub"] = sub(sym, step_lat).astype(np.int32)
    o["sym_hist"] = np.bincount((sym.reshape(-1).numpy() + 256).clip(0, 512), minlength=513)
    o["y_hat_sub"] = sub(y_hat, step_lat)
    if full_ints:
        o["idx_full"] = idx.reshape(-1).numpy().astype(np.int8)
        o["sym_full"] = sym.reshape(-1).numpy().astype(np.int16)
        assert np.array_equal(o["sym_full"].astype(np.int64), sym.reshape(-1).numpy().astype(np.int64))
        o.update(margins(net, y, z, scales, means))
    o["bits_y"] = np.array([float((-torch.log2(y_lik)).sum())])
    o["bits_z"] = np.array([float((-torch.log2(z_lik)).sum())])
    t0 = time.time()
    y_strings = net.gaussian_conditional.compress(y, idx, means=means)
    print(tag, "rans y", time.time() - t0, len(y_strings[0]))
    o["y_string_len"] = np.array([len(y_strings[0])])
    o["z_string"] = np.frombuffer(z_strings[0], dtype=np.uint8)
    if len(y_strings[0]) < 400000:
        o["y_string"] = np.frombuffer(y_strings[0], dtype=np.uint8)
    import hashlib
    o["y_string_sha256"] = np.frombuffer(hashlib.sha256(y_strings[0]).digest(), dtype=np.uint8)
    # hyper-decoder given a synthetic, regenerable z_hat (independent of round() flips of z)
    zs = synth_zhat(z.shape[1], 6) + net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)
    p2 = net.h_s(zs)
    o["hs_synth_sub"], o["hs_synth_stats"] = sub(p2, step_lat), stats(p2)
    o["hs_synth_idx_hist"] = np.bincount(
        net.gaussian_conditional.build_indexes(p2.chunk(2, 1)[0]).reshape(-1).numpy(), minlength=64)
    # decoder given a synthetic, regenerable y_hat (independent of round() flips)
    t0 = time.time()
    x_hat = net.decode_latent(yhat_synth)
    print(tag, "g_s", time.time() - t0)
    o["xhat_sub"], o["xhat_stats"] = sub(x_hat, step_img), stats(x_hat)
    # overlap rows (10, 20, ...) get two contributions: keep one full overlap row
    o["xhat_row10_c0"] = x_hat[0, 0, 10].numpy()
    o["xhat_row720_c0"] = x_hat[0, 0, 720].numpy()
    return o


def synth_yhat(latent, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.round(2.0 * torch.randn(1, latent, 72, 144, generator=g)) + torch.randn(1, latent, 72, 144, generator=g)


def synth_zhat(cz, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.round(3.0 * torch.randn(1, cz, 18, 36, generator=g))


def margins(net, y, z, scales, means):
    """How far the reference's own values sit from the nearest rounding / table boundary: the float error another
    implementation may have on z, y - mu and sigma before ONE integer of the frame differs."""
    med = net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)

    def half_dist(v):
        v = v.double()
        return float((v - torch.floor(v) - 0.5).abs().min())
    table = net.gaussian_conditional.scale_table.double()
    sg = scales.double().reshape(-1, 1)
    m_s = float((sg - table[:-1].reshape(1, -1)).abs().min())
    return {"margin_z": np.array([half_dist(z - med)]), "margin_y": np.array([half_dist(y - means)]),
            "margin_scale": np.array([m_s])}


def stage_thin_search(seeds=range(100, 164)):
    """Pick the input seed of the SECOND thin fixture (thin_e2e_b.npz).  round() and the scale-table search make the
    integer side discontinuous: another implementation agrees with the reference on EVERY z symbol, CDF index and y
    symbol of a frame only if its float error stays below the frame's margins (`margins`).  With 165 888 latents the
    smallest margin of a random frame is ~1e-6 - the size of fp32 noise itself (thin_e2e.npz's frame: the product
    flips 1 index).  This stage runs the reference on 64 candidate frames and prints their margins; the fixture
    uses the seed with the widest one (`THIN_B_SEED`), so that byte equality with the reference-written stream and
    cross-implementation decode can be asserted instead of x-failed."""
    net = build_thin()
    load_synth(net, seed=7)
    best = None
    for sd in seeds:
        x = synth.synth_frame(8, seed=sd).unsqueeze(0)
        moments = net.quant_conv(net.g_a(x))
        y = moments[:, : moments.shape[1] // 2]
        z = net.h_a(y)
        z_hat, _ = net.entropy_bottleneck(z)
        scales, means = net.h_s(z_hat).chunk(2, 1)
        m = margins(net, y, z, scales, means)
        worst = min(float(m["margin_z"][0]) / 4, float(m["margin_y"][0]), float(m["margin_scale"][0]))
        print(f"seed {sd}: margin z {m['margin_z'][0]:.2e}  y {m['margin_y'][0]:.2e}  scale {m['margin_scale'][0]:.2e}"
              f"  -> score {worst:.2e}", flush=True)
        if best is None or worst > best[0]:
            best = (worst, sd)
    print("best seed", best[1], "score", best[0])


def stage_thin_cands(seeds=None):
    """Scratch fixtures (tests/golden/_cand/, git-ignored) with the reference's full integer side for a few candidate
    seeds: tools/thin_seed_probe.py counts the product's disagreements with each on the GPU box."""
    seeds = seeds or [int(v) for v in os.environ.get("THIN_CANDS", "").split(",") if v]
    net = build_thin()
    load_synth(net, seed=7)
    os.makedirs(os.path.join(HERE, "_cand"), exist_ok=True)
    for sd in seeds:
        x = synth.synth_frame(8, seed=sd).unsqueeze(0)
        moments = net.quant_conv(net.g_a(x))
        y = moments[:, : moments.shape[1] // 2]
        z = net.h_a(y)
        med = net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)
        z_sym = net.entropy_bottleneck.quantize(z, "symbols", med)
        z_hat, _ = net.entropy_bottleneck(z)
        scales, means = net.h_s(z_hat).chunk(2, 1)
        idx = net.gaussian_conditional.build_indexes(scales)
        sym = net.gaussian_conditional.quantize(y, "symbols", means)
        np.savez_compressed(os.path.join(HERE, "_cand", f"thin_cand_{sd}.npz"), z_sym=z_sym.numpy().astype(np.int32),
                            idx_full=idx.reshape(-1).numpy().astype(np.int8),
                            sym_full=sym.reshape(-1).numpy().astype(np.int16), **margins(net, y, z, scales, means))
        print("cand", sd, flush=True)


THIN_B_SEED = 162   # widest margins of the 64 candidates (z 2.3e-4, y 4.5e-6, scale 2.1e-6); product: 0 flips (tools/thin_seed_probe.py, round 3: 10 of 16 candidates had none)


def stage_thin2():
    net = build_thin()
    load_synth(net, seed=7)
    x = synth.synth_frame(8, seed=THIN_B_SEED).unsqueeze(0)
    o = run_e2e(net, x, synth_yhat(16, 5), step_lat=37, step_img=1009, tag="thin_b", full_ints=True)
    o["x_seed"] = np.array([THIN_B_SEED])
    for k in ("xhat_sub", "xhat_stats", "xhat_row10_c0", "xhat_row720_c0", "hs_synth_sub", "hs_synth_stats",
              "hs_synth_idx_hist"):
        o.pop(k, None)       # input-independent: already in thin_e2e.npz
    # the reference's own decode of its own stream (x_hat from the .bin): what a cross-implementation decode must hit
    out = net.compress(x)
    assert out["strings"][0][0] == o["y_string"].tobytes() and out["strings"][1][0] == o["z_string"].tobytes()
    rec = net.decompress(out["strings"], out["z_shape"])
    o["xhat_rt_sub"] = sub(rec["x_hat"], 1009)
    np.savez_compressed(os.path.join(HERE, "thin_e2e_b.npz"), **o)
    print("thin2 done")


THIN_A_SEED = 135   # round 5 (was 2, the documented one-index-flip frame): reference margins z 7.0e-5, y 9.5e-6, scale 1.4e-6;
                    # product: no flip (tools/thin_seed_probe.py over 36 candidate frames: 20 agree on every integer)


def stage_thin():
    net = build_thin()
    load_synth(net, seed=7)
    x = synth.synth_frame(8, seed=THIN_A_SEED).unsqueeze(0)
    # (round 4: full_ints - every CDF index / y symbol of the reference run, so that the documented flip case of this
    # frame is pinned element-wise: exactly which index differs, and the reference's integers through the product coder)
    o = run_e2e(net, x, synth_yhat(16, 5), step_lat=37, step_img=1009, tag="thin", full_ints=True)
    o["x_seed"] = np.array([THIN_A_SEED])
    np.savez_compressed(os.path.join(HERE, "thin_e2e.npz"), **o)
    print("thin done")


THIN_FLIP_SEED = 2   # rounds 1-4's frame a: the product's h_s rounds ONE of its 165 888 scales into the neighbouring table row
                     # (within 1e-5 of the boundary) - the documented cross-implementation flip case, kept as a fixture of
                     # its own (round 6) beside the agreeing frames a (seed 135) and b (seed 162)


def stage_thin_flip():
    net = build_thin()
    load_synth(net, seed=7)
    x = synth.synth_frame(8, seed=THIN_FLIP_SEED).unsqueeze(0)
    o = run_e2e(net, x, synth_yhat(16, 5), step_lat=37, step_img=1009, tag="thin_flip", full_ints=True)
    o["x_seed"] = np.array([THIN_FLIP_SEED])
    np.savez_compressed(os.path.join(HERE, "thin_e2e_flip.npz"), **o)
    print("thin_flip done")


def stage_thin_cands_pack():
    """tests/golden/_cand/thin_cand_*.npz (scratch, stage thin_cands) -> ONE committed fixture thin_cands.npz: the
    reference's integers of 36 thin frames (z symbols, CDF indexes, y symbols), from which the test rebuilds the streams
    the reference's compress() writes (coder on the reference's integers == reference-python-written stream, pinned in
    tests/test_reference_streams.py) and checks which of them decode on the product and which must be refused."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "_cand", "thin_cand_*.npz")),
                   key=lambda f: int(os.path.basename(f)[len("thin_cand_"):-4]))
    seeds, z, idx, sym = [], [], [], []
    for f in files:
        g = np.load(f)
        seeds.append(int(os.path.basename(f)[len("thin_cand_"):-4]))
        assert np.abs(g["z_sym"]).max() < 32768
        z.append(g["z_sym"].reshape(-1).astype(np.int16))
        idx.append(g["idx_full"].astype(np.int8))
        sym.append(g["sym_full"].astype(np.int16))
    np.savez_compressed(os.path.join(HERE, "thin_cands.npz"), seeds=np.array(seeds), z_sym=np.stack(z),
                        idx_full=np.stack(idx), sym_full=np.stack(sym))
    print("packed", len(seeds), "candidate frames")


def stage_full():
    net = VAEformer(268).eval()
    load_synth(net, seed=7)
    x = synth.synth_frame(268, seed=2).unsqueeze(0)
    o = run_e2e(net, x, synth_yhat(256, 5), step_lat=499, step_img=99991, tag="full", full_ints=True)
    # round 4: every integer of the reference's frame (z symbols, CDF indexes, y symbols) in a file of its own, so
    # that the full-size comparisons no longer depend on the product reproducing all 165 888 z symbols: the test
    # injects the reference's z_hat into the product's h_s and the reference's (index, symbol) pairs into its coder
    ints = {k: o.pop(k) for k in ("z_sym_full", "idx_full", "sym_full")}
    ints["y_string_sha256"], ints["y_string_len"], ints["z_string"] = o["y_string_sha256"], o["y_string_len"], o["z_string"]
    np.savez_compressed(os.path.join(HERE, "full268.npz"), **o)
    np.savez_compressed(os.path.join(HERE, "full268_ints.npz"), **ints)
    print("full done")


def latent_ints(net, y, tag):
    """Every integer of the reference's latent side for one frame's y, + the streams its compress() writes
    (full268_ints-style): z symbols, CDF indexes, y symbols, stream hashes / bytes, margins, sub-sampled sigma / mu."""
    import hashlib
    o = {}
    z = net.h_a(y)
    med = net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)
    z_sym = net.entropy_bottleneck.quantize(z, "symbols", med)
    z_hat, _ = net.entropy_bottleneck(z)
    scales, means = net.h_s(z_hat).chunk(2, 1)
    idx = net.gaussian_conditional.build_indexes(scales)
    sym = net.gaussian_conditional.quantize(y, "symbols", means)
    assert int(z_sym.abs().max()) < 32768 and int(sym.abs().max()) < 32768 and int(idx.max()) < 128
    o["z_sym_full"] = z_sym.reshape(-1).numpy().astype(np.int16)
    o["idx_full"] = idx.reshape(-1).numpy().astype(np.int8)
    o["sym_full"] = sym.reshape(-1).numpy().astype(np.int16)
    o.update(margins(net, y, z, scales, means))
    step = 499 if y.numel() > 1000000 else 37
    o["scales_sub"], o["means_sub"] = sub(scales, step), sub(means, step)
    o["y_sub"], o["y_stats"] = sub(y, step), stats(y)
    z_strings = net.entropy_bottleneck.compress(z)
    y_strings = net.gaussian_conditional.compress(y, idx, means=means)
    o["z_string"] = np.frombuffer(z_strings[0], dtype=np.uint8)
    o["y_string_len"] = np.array([len(y_strings[0])])
    o["y_string_sha256"] = np.frombuffer(hashlib.sha256(y_strings[0]).digest(), dtype=np.uint8)
    if len(y_strings[0]) < 400000:
        o["y_string"] = np.frombuffer(y_strings[0], dtype=np.uint8)
    gc = net.gaussian_conditional
    v = sym.reshape(-1) - gc._offset[idx.reshape(-1)]
    o["n_escape"] = np.array([int(((v < 0) | (v >= gc._cdf_length[idx.reshape(-1)] - 2)).sum())])
    print(tag, "y bytes", len(y_strings[0]), "z bytes", len(z_strings[0]), "escapes", int(o["n_escape"][0]), flush=True)
    return o


def _swap_variant(net, variant):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if k in synth._MATCHED_KEYS}
    sd = synth.fill_state_dict(shapes, 7, variant=variant)
    torch.nn.Module.load_state_dict(net, sd, strict=False)


def stage_ints(seeds=None):
    """Round 5: reference integers + stream hashes (a) of the 268 fixture frame (seed 2) under the entropy-matched weight
    variant (cra5_amd/synth.py MATCHED_SIGMA) -> full268_m_ints.npz, and (b) of the first two frames of the BENCHMARKED set
    (bench.py: x_f = synth_frame(268, 1000 + f), f = 0, 1) under both variants -> bench{seed}_ints.npz /
    bench{seed}_m_ints.npz.  One g_a pass per frame (~2-3 min each), the latent side once per variant."""
    net = VAEformer(268).eval()
    load_synth(net, seed=7)
    for sd_x, name in ((2, "full268"), (1000, "bench1000"), (1001, "bench1001")):
        if seeds and sd_x not in seeds:
            continue
        x = synth.synth_frame(268, seed=sd_x).unsqueeze(0)
        t0 = time.time()
        moments = net.quant_conv(net.g_a(x))
        y = moments[:, : moments.shape[1] // 2].contiguous()
        print(name, "g_a", time.time() - t0, flush=True)
        for variant, suffix in (("default", ""), ("matched", "_m")):
            if name == "full268" and variant == "default":
                continue                       # full268_ints.npz exists (stage_full)
            _swap_variant(net, variant)
            o = latent_ints(net, y, name + suffix)
            o["x_seed"] = np.array([sd_x])
            np.savez_compressed(os.path.join(HERE, f"{name}{suffix}_ints.npz"), **o)
        _swap_variant(net, "default")
    print("ints done")


def stage_thin_matched(seed=THIN_B_SEED):
    """The thin model under the entropy-matched weight variant: every integer, the streams, the reference's own round trip."""
    net = build_thin()
    load_synth(net, seed=7, variant="matched")
    x = synth.synth_frame(8, seed=seed).unsqueeze(0)
    moments = net.quant_conv(net.g_a(x))
    y = moments[:, : moments.shape[1] // 2].contiguous()
    o = latent_ints(net, y, "thin_m")
    o["x_seed"] = np.array([seed])
    out = net.compress(x)
    assert out["strings"][0][0] == o["y_string"].tobytes() and out["strings"][1][0] == o["z_string"].tobytes()
    rec = net.decompress(out["strings"], out["z_shape"])
    o["xhat_rt_sub"] = sub(rec["x_hat"], 1009)
    np.savez_compressed(os.path.join(HERE, "thin_e2e_m.npz"), **o)
    print("thin matched done")


def build_159():
    """The 268 architecture with 159 variables through the reference's ddconfig route
    (model_version != 268): same kwargs as the hard-wired 268 config, in_chans = out_chans = 159."""
    dd_kw = dict(z_dim=None, learnable_pos=True, window=True, window_size=[(24, 24), (12, 48), (48, 12)],
                 interval=4, drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                 test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=True, img_size=(721, 1440))
    pr_kw = dict(z_dim=256, embed_dim=360, depth=8, num_heads=5, interval=1, learnable_pos=True, window=False,
                 drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                 test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=False, img_size=(72, 144))
    return VAEformer(0, embed_dim=256, z_channels=256, y_channels=1024, sample_posterior=False,
                     frozen_encoder=False, lower_dim=True,
                     ddconfig=dict(arch='vit_large', pretrained_model='', patch_size=(11, 10), patch_stride=(10, 10),
                                   in_chans=159, out_chans=159, kwargs=dd_kw),
                     priorconfig=dict(pretrained_model='', patch_size=(4, 4), in_chans=256, out_chans=256,
                                      kwargs=pr_kw)).eval()


def stage_full159():
    """BASELINE configs[1]: x ~ N(0,1) seed 1, encode_to_latent + latent_to_reconstruction."""
    net = build_159()
    shapes = load_synth(net, seed=3)
    x = synth.synth_frame(159, seed=1).unsqueeze(0)
    o = {}
    t0 = time.time()
    y, _, _ = net.encode_latent(x, type='float')
    print("159 g_a", time.time() - t0)
    o["y_sub"], o["y_stats"] = sub(y, 499), stats(y)
    t0 = time.time()
    x_hat = net.decode_latent(synth_yhat(256, 5))
    print("159 g_s", time.time() - t0)
    o["xhat_sub"], o["xhat_stats"] = sub(x_hat, 99991), stats(x_hat)
    o["xhat_row10_c0"] = x_hat[0, 0, 10].numpy()
    o["xhat_row720_c158"] = x_hat[0, 158, 720].numpy()
    # end to end on the reference's own quantised latent: x_hat(y_hat(x)) sub-sampled
    _, y_hat, _ = net.encode_latent(x, type='quantized')
    o["y_hat_sub"] = sub(y_hat, 499)
    # the integer side of that quantised latent (round 3): with z symbols and CDF indexes on record a consumer can tell
    # a rounding flip of z (which moves every mu by ~1e-3 through h_s) from an error of its h_s
    z = net.h_a(y)
    o["z_sub"], o["z_stats"] = sub(z, 13), stats(z)
    med = net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)
    z_sym = net.entropy_bottleneck.quantize(z, "symbols", med).numpy().astype(np.int32)
    o["z_sym_hist"] = np.bincount((z_sym.reshape(-1) + 64).clip(0, 128), minlength=129)
    o["z_sym"] = z_sym.reshape(-1).astype(np.int16)
    z_hat, _ = net.entropy_bottleneck(z)
    scales, means = net.h_s(z_hat).chunk(2, 1)
    o["scales_sub"], o["means_sub"] = sub(scales, 499), sub(means, 499)
    idx = net.gaussian_conditional.build_indexes(scales)
    o["idx_sub"] = sub(idx, 499).astype(np.int32)
    o["idx_hist"] = np.bincount(idx.reshape(-1).numpy(), minlength=64)
    sym = net.gaussian_conditional.quantize(y, "symbols", means)
    o["sym_sub"] = sub(sym, 499).astype(np.int32)
    o["sym_hist"] = np.bincount((sym.reshape(-1).numpy() + 256).clip(0, 512), minlength=513)
    assert torch.equal(y_hat, sym.float() + means)
    o.update(margins(net, y, z, scales, means))
    np.savez_compressed(os.path.join(HERE, "full159.npz"), **o)
    sk = json.load(open(os.path.join(HERE, "state_keys.json")))
    sk["v159"] = {k: list(v) for k, v in shapes.items()}
    json.dump(sk, open(os.path.join(HERE, "state_keys.json"), "w"))
    print("full159 done")


def stage_cnn():
    """the reference's CNN zoo classes (models/google.py:64-508) at N=32, M=48 on one
    3 x 128 x 192 image, synthetic weights from cra5_amd/synth.py -> cnn_zoo.npz."""
    from cra5.models.compressai.models.google import FactorizedPrior, MeanScaleHyperprior, ScaleHyperprior
    N, M = 32, 48
    g = torch.Generator().manual_seed(77)
    x = torch.rand(1, 3, 128, 192, generator=g)
    o, keys = {"x": x.numpy()}, {}
    for name, cls in (("factorized", FactorizedPrior), ("hyperprior", ScaleHyperprior), ("meanscale", MeanScaleHyperprior)):
        net = cls(N, M).eval()
        shapes = load_synth(net, seed=11)
        keys[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
        y = net.g_a(x)
        o[f"{name}_y"] = y.numpy()
        fw = net(x)
        o[f"{name}_xhat_fw"] = sub(fw["x_hat"], 5)
        for k, v in fw["likelihoods"].items():
            o[f"{name}_bits_{k}"] = np.array([float((-torch.log2(v)).sum())])
        gy = torch.Generator().manual_seed(5)
        y_hat = torch.round(3.0 * torch.randn(y.shape, generator=gy))
        o[f"{name}_xhat_synth"] = sub(net.g_s(y_hat), 5)                 # decoder on a regenerable y_hat
        out = net.compress(x)
        rec = net.decompress(out["strings"], out["shape"])
        o[f"{name}_xhat_rt"] = sub(rec["x_hat"], 5)
        for i, ss in enumerate(out["strings"]):
            o[f"{name}_string{i}"] = np.frombuffer(ss[0], dtype=np.uint8)
        o[f"{name}_shape"] = np.array(list(out["shape"]))
        if name != "factorized":
            z = net.h_a(y if name == "meanscale" else torch.abs(y))
            o[f"{name}_z"] = z.numpy()
            z_hat, _ = net.entropy_bottleneck(z)
            p = net.h_s(z_hat)
            o[f"{name}_hs"] = p.numpy()
    np.savez_compressed(os.path.join(HERE, "cnn_zoo.npz"), **o)
    sk = json.load(open(os.path.join(HERE, "state_keys.json")))
    sk["cnn"] = keys
    json.dump(sk, open(os.path.join(HERE, "state_keys.json"), "w"))
    print("cnn done")


def stage_cnn_relu():
    """FactorizedPriorReLU (models/google.py:166-199) like stage_cnn -> cnn_relu.npz.  As shipped the class cannot be
    constructed: its __init__ ends in `MODELS.build(rate_distortion_loss)` (a training criterion) and the module never
    imports a `MODELS` (NameError); the codec path never touches the criterion, so the name is bound to a registry
    whose build() returns None."""
    import types
    from cra5.models.compressai.models import google
    N, M = 32, 48
    google.MODELS = types.SimpleNamespace(build=lambda cfg: None)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(1, 3, 128, 192, generator=g)
    o = {}
    net = google.FactorizedPriorReLU(N, M, None).eval()
    load_synth(net, seed=11)
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    y = net.g_a(x)
    o["y"] = y.numpy()
    fw = net(x)
    o["xhat_fw"] = sub(fw["x_hat"], 5)
    o["bits_y"] = np.array([float((-torch.log2(fw["likelihoods"]["y"])).sum())])
    gy = torch.Generator().manual_seed(5)
    y_hat = torch.round(3.0 * torch.randn(y.shape, generator=gy))
    o["xhat_synth"] = sub(net.g_s(y_hat), 5)
    out = net.compress(x)
    o["xhat_rt"] = sub(net.decompress(out["strings"], out["shape"])["x_hat"], 5)
    o["string0"] = np.frombuffer(out["strings"][0][0], dtype=np.uint8)
    o["shape"] = np.array(list(out["shape"]))
    np.savez_compressed(os.path.join(HERE, "cnn_relu.npz"), **o)
    sk = json.load(open(os.path.join(HERE, "state_keys.json")))
    sk["cnn"]["factorized_relu"] = keys
    json.dump(sk, open(os.path.join(HERE, "state_keys.json"), "w"))
    print("cnn_relu done")


def stage_stats():
    """The reference's cra5_api.get_mean_std / channel_vname_mapping run on the reference's own
    config + JSON files (the methods only read self.cfg / self.level_mapping, so they are called
    on a bare object: the constructor needs the network)."""
    import importlib
    import types

    class Config(dict):
        """Import stand-in for cra5.utils.config.Config (needs yapf / addict, absent here): the
        reference's config file is plain Python, exec'd into an attribute dict."""
        __getattr__ = dict.__getitem__

        @staticmethod
        def fromfile(path):
            ns = {}
            exec(compile(open(path).read(), path, "exec"), ns)  # noqa: S102
            return Config({k: v for k, v in ns.items() if not k.startswith("__")})

    stubs = {"cra5.utils.config": dict(Config=Config), "cra5.api.era5_downloader": dict(era5_downloader=object)}
    for name in ("matplotlib", "matplotlib.pyplot", "xarray", "cdsapi"):
        try:
            importlib.import_module(name)
        except Exception:  # noqa: BLE001
            stubs[name] = {}
    for name, attrs in stubs.items():
        m = types.ModuleType(name)
        m.__path__ = []
        m.__dict__.update(attrs)
        sys.modules[name] = m
    import cra5.api  # noqa: F401
    ref_api_mod = sys.modules['cra5.api.cra5_api']
    api = object.__new__(ref_api_mod.cra5_api)
    api.cfg = Config.fromfile(os.path.join(os.path.dirname(ref_api_mod.__file__), "cra5_268v_config.py"))
    api.level_mapping = [api.cfg.total_levels.index(v) for v in api.cfg.pressure_level if v in api.cfg.total_levels]
    mean, std = api.get_mean_std()
    c2v, v2c = api.channel_vname_mapping()
    assert mean.shape == (268,) and len(c2v) == 268
    np.savez_compressed(os.path.join(HERE, "era5_stats_ref.npz"), mean=mean, std=std,
                        vnames=np.array([c2v[i] for i in range(268)]),
                        level_mapping=np.array(api.level_mapping, dtype=np.int64))
    print("stats done")


def stage_fp64(which="full"):
    """The reference in float64 on the same weights / inputs: calibrates the fp32 noise floor
    (how far the reference's OWN fp32 path is from exact arithmetic).  ~6 min, ~40 GB for
    the 268 model."""
    if which == "full":
        net, cin, lat, sl, si = VAEformer(268).eval(), 268, 256, 499, 99991
    else:
        net, cin, lat, sl, si = build_thin(), 8, 16, 37, 1009
    load_synth(net, seed=7)
    net = net.double()
    x = synth.synth_frame(cin, seed=2 if which == "full" else THIN_A_SEED).unsqueeze(0).double()
    o = {}
    t0 = time.time()
    moments = net.quant_conv(net.g_a(x))
    y = moments[:, : moments.shape[1] // 2]
    print(which, "fp64 g_a", time.time() - t0)
    o["y_sub"] = sub(y, sl)
    z = net.h_a(y)
    o["z_sub"] = sub(z, 13)
    zs = synth_zhat(z.shape[1], 6).double() + net.entropy_bottleneck._get_medians().reshape(1, -1, 1, 1)
    o["hs_synth_sub"] = sub(net.h_s(zs), sl)
    t0 = time.time()
    x_hat = net.decode_latent(synth_yhat(lat, 5).double())
    print(which, "fp64 g_s", time.time() - t0)
    o["xhat_sub"] = sub(x_hat, si)
    np.savez_compressed(os.path.join(HERE, f"{'full268' if which == 'full' else 'thin'}_fp64.npz"), **o)
    print(which, "fp64 done")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", nargs="+", default=["small", "thin"])
    a = ap.parse_args()
    for s in a.stage:
        dict(small=stage_small, thin=stage_thin, thin_flip=stage_thin_flip, thin_cands_pack=stage_thin_cands_pack, thin_search=stage_thin_search, thin_cands=stage_thin_cands, thin2=stage_thin2, full=stage_full, full159=stage_full159, stats=stage_stats,
             cnn=stage_cnn, cnn_relu=stage_cnn_relu, ints=stage_ints, thin_matched=stage_thin_matched,
             thin64=lambda: stage_fp64("thin"), full64=lambda: stage_fp64("full"))[s]()
