"""Pins the CPU oracle (oracle/) against golden vectors produced by the REFERENCE's own
Python / C++ (tests/golden/make_golden.py, run in the build container where
/root/reference exists).  Runs on CPU in seconds; nothing here touches /root/reference."""
import json

import numpy as np
import pytest
import torch

from cra5_amd import synth
from oracle import cbind, rans_py
from oracle import torch_ref as R


def rmse(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return float(torch.sqrt(torch.mean((a - b) ** 2)))


def sub(t, step):
    return t.detach().reshape(-1)[::step]


def test_pmf_to_cdf_matches_reference_cxx(golden_dir):
    """oracle C restatement == the reference's ops.cpp outputs, bit-exact."""
    g = json.load(open(f"{golden_dir}/pmf_cdf.json"))
    for c in g["cases"]:
        got = cbind.pmf_to_cdf(np.array(c["pmf"], dtype=np.float32), c["precision"])
        assert got.tolist() == c["cdf"]
    for e in g["errors"]:
        vals = [float(v) for v in e["pmf"]]
        if e["raises"]:
            with pytest.raises(ValueError):
                cbind.pmf_to_cdf(np.array(vals, dtype=np.float32))


def test_live_reference_cxx_if_built():
    """When oracle/_ref/_CXX (the reference's ops.cpp compiled where it lies) is present,
    cross-check on fresh random pmfs as well."""
    cxx = cbind.ref_cxx()
    if cxx is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(7)
    for _ in range(200):
        n = int(rng.integers(1, 300))
        p = rng.random(n).astype(np.float32) ** int(rng.integers(1, 10))
        p /= max(p.sum(), 1e-30)
        if p.sum() == 0:
            continue
        assert cbind.pmf_to_cdf(p).tolist() == list(cxx.pmf_to_quantized_cdf(p.tolist(), 16))


def test_tables_match_reference(golden_dir):
    g = np.load(f"{golden_dir}/tables_default.npz")
    cdf, ln, off = R.gc_tables(R.get_scale_table(), cbind.pmf_to_cdf)
    assert np.array_equal(g["scale_table"], R.get_scale_table().numpy())
    assert np.array_equal(g["gc_cdf"], cdf.numpy()) and np.array_equal(g["gc_len"], ln.numpy())
    assert np.array_equal(g["gc_off"], off.numpy())
    assert cdf.shape == (64, 3133)
    keys = json.load(open(f"{golden_dir}/state_keys.json"))["thin"]
    sd = synth.fill_state_dict({k: tuple(v) for k, v in keys.items()}, seed=7)
    cdf, ln, off = R.eb_tables(sd, cbind.pmf_to_cdf)
    assert np.array_equal(g["eb_cdf"], cdf.numpy()) and np.array_equal(g["eb_len"], ln.numpy())
    assert np.array_equal(g["eb_off"], off.numpy())


def test_ops_match_reference(golden_dir):
    """Window attention (incl. the (48,12) zero-pad semantics), global attention at head dim
    64 / 72, whole blocks: oracle restatement vs the reference modules."""
    g = np.load(f"{golden_dir}/ops_small.npz")
    H, W, C, heads = 72, 144, 128, 2
    gen = torch.Generator().manual_seed(11)
    xtok = torch.randn(1, H * W, C, generator=gen)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C),
              "attn.proj.bias": (C,)}
    sd = synth.fill_state_dict(shapes, seed=21)
    for name, ws in (("w24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
        out = R.attention_window(xtok, sd, "attn", heads, H, W, ws)
        assert rmse(sub(out, 61), g[f"winattn_{name}"]) < 1e-6
    sd72 = synth.fill_state_dict({"attn.qkv.weight": (432, 144), "attn.qkv.bias": (432,),
                                  "attn.proj.weight": (144, 144), "attn.proj.bias": (144,)}, seed=22)
    x648 = torch.randn(1, 648, 144, generator=gen)
    assert rmse(sub(R.attention_global(x648, sd72, "attn", 2), 7), g["globattn_hd72"]) < 1e-6
    sd64 = synth.fill_state_dict(shapes, seed=23)
    assert rmse(sub(R.attention_global(xtok, sd64, "attn", heads), 61), g["globattn_hd64"]) < 1e-6
    bshapes = {"blocks.0.norm1.weight": (C,), "blocks.0.norm1.bias": (C,), "blocks.0.attn.qkv.weight": (3 * C, C),
               "blocks.0.attn.qkv.bias": (3 * C,), "blocks.0.attn.proj.weight": (C, C),
               "blocks.0.attn.proj.bias": (C,), "blocks.0.norm2.weight": (C,), "blocks.0.norm2.bias": (C,),
               "blocks.0.mlp.fc1.weight": (4 * C, C), "blocks.0.mlp.fc1.bias": (4 * C,),
               "blocks.0.mlp.fc2.weight": (C, 4 * C), "blocks.0.mlp.fc2.bias": (C,)}
    bsd = synth.fill_state_dict(bshapes, seed=24)
    assert rmse(sub(R.block(xtok, bsd, "blocks.0", heads, H, W, (48, 12)), 61), g["blk_w48x12"]) < 2e-6
    assert rmse(sub(R.block(xtok, bsd, "blocks.0", heads, H, W, None), 61), g["blk_glob"]) < 2e-6
    for inv in (0, 1):
        y = R.gdn(torch.from_numpy(g[f"gdn_inv{inv}_x"]), torch.from_numpy(g[f"gdn_inv{inv}_beta"]),
                  torch.from_numpy(g[f"gdn_inv{inv}_gamma"]), inverse=bool(inv))
        assert rmse(y, g[f"gdn_inv{inv}_y"]) < 1e-6


@pytest.fixture(scope="module")
def thin_oracle(golden_dir):
    keys = json.load(open(f"{golden_dir}/state_keys.json"))["thin"]
    sd = synth.fill_state_dict({k: tuple(v) for k, v in keys.items()}, seed=7)
    cfg = R.cfg_thin()
    tb = R.tables(sd, cbind.pmf_to_cdf)
    x = synth.synth_frame(8, seed=int(np.load(f"{golden_dir}/thin_e2e.npz")["x_seed"][0])).unsqueeze(0)
    with torch.no_grad():
        out, y = R.compress(x, sd, cfg, tb, cbind.rans_encode)
        side = R.latent_side(y, sd, cfg, tb["scale_table"])
    return sd, cfg, tb, x, y, side, out


def test_thin_path_matches_reference(thin_oracle, golden_dir):
    """Whole encode side of the full-spatial thin model: every stage of the oracle vs the
    reference's outputs, INCLUDING the byte streams the reference's own compress() python
    produced (with the oracle coder plugged in as `compressai.ans`)."""
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    sd, cfg, tb, x, y, side, out = thin_oracle
    assert rmse(sub(y, 37), g["y_sub"]) < 1e-6
    assert rmse(side["z"], g["z"]) < 1e-6
    assert np.array_equal(side["z_sym"].numpy(), g["z_sym"])
    assert rmse(sub(side["scales"], 37), g["scales_sub"]) < 1e-6
    assert rmse(sub(side["means"], 37), g["means_sub"]) < 1e-6
    assert np.array_equal(sub(side["idx"], 37).numpy(), g["idx_sub"])
    assert np.array_equal(sub(side["y_sym"], 37).numpy(), g["sym_sub"])
    assert np.array_equal(np.bincount(side["idx"].reshape(-1).numpy(), minlength=64), g["idx_hist"])
    assert out["strings"][1][0] == g["z_string"].tobytes()
    assert out["strings"][0][0] == g["y_string"].tobytes()
    assert tuple(out["z_shape"]) == (18, 36)
    _, y_lik = R.gc_forward(y, side["scales"], side["means"])
    assert rmse(sub(y_lik, 37), g["y_lik_sub"]) < 1e-7
    _, z_lik = R.eb_forward(side["z"], sd)
    assert rmse(sub(z_lik, 13), g["z_lik_sub"]) < 1e-7


def test_thin_decode_matches_reference(thin_oracle, golden_dir):
    g = np.load(f"{golden_dir}/thin_e2e.npz")
    sd, cfg, tb, x, y, side, out = thin_oracle
    gen = torch.Generator().manual_seed(5)
    yh = torch.round(2.0 * torch.randn(1, 16, 72, 144, generator=gen)) + torch.randn(1, 16, 72, 144, generator=gen)
    with torch.no_grad():
        x_hat = R.g_s(yh, sd, cfg)
        assert rmse(sub(x_hat, 1009), g["xhat_sub"]) < 1e-6
        assert rmse(x_hat[0, 0, 10], g["xhat_row10_c0"]) < 1e-6
        assert rmse(x_hat[0, 0, 720], g["xhat_row720_c0"]) < 1e-6
        gz = torch.Generator().manual_seed(6)
        zs = torch.round(3.0 * torch.randn(1, 16, 18, 36, generator=gz)) + R.eb_medians(sd).reshape(1, -1, 1, 1)
        assert rmse(sub(R.h_s(zs, sd, cfg), 37), g["hs_synth_sub"]) < 1e-6
        # stream round trip through the oracle decoder
        y_hat = R.decompress(out["strings"], out["z_shape"], sd, cfg, tb, cbind.rans_decode, "latent")
        assert torch.equal(y_hat, side["y_hat"])


def test_fp32_noise_floor_calibration(golden_dir):
    """How far the REFERENCE's own fp32 path is from float64 on these synthetic weights -
    the yardstick the GPU tolerances are read against."""
    import os
    for name, s_lat, s_img in (("thin", 37, 1009), ("full268", 499, 99991)):
        p64 = f"{golden_dir}/{name if name == 'full268' else 'thin'}_fp64.npz"
        p32 = f"{golden_dir}/{'full268' if name == 'full268' else 'thin_e2e'}.npz"
        if not os.path.exists(p64):
            pytest.skip("fp64 calibration fixture not generated")
        a, b = np.load(p32), np.load(p64)
        e_y, e_x = rmse(a["y_sub"], b["y_sub"]), rmse(a["xhat_sub"], b["xhat_sub"])
        e_h = rmse(a["hs_synth_sub"], b["hs_synth_sub"])
        print(f"{name}: reference fp32 vs fp64: y {e_y:.2e}, hs {e_h:.2e}, x_hat {e_x:.2e}")
        assert e_y < 5e-5 and e_x < 5e-5
