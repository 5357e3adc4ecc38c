"""Deterministic synthetic weights.

There is no network on the build / GPU boxes, so neither the reference checkpoint
(`cra5_268v_300k.pth`, zoo/image.py:73 in the reference) nor ERA5 samples exist.
`fill_state_dict` produces closed-form, seed-reproducible values for every
parameter of a VAEformer state dict (reference key names), scaled so that the
entropy-coding side is exercised (scale indexes spread over the table, non-zero
symbols, a few escape symbols).  bench.py, the tests and the golden-vector generator
all call this one function, so the same weights can be rebuilt anywhere without
shipping tensors.
"""
import zlib

import numpy as np
import torch

_SKIP = ("._offset", "._quantized_cdf", "._cdf_length", ".scale_table", ".scale_bound", ".bound", ".target")


def _rng(seed, key):
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(key.encode())])


def _randn(rng, shape, std):
    return torch.from_numpy((rng.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(std)))


# "matched" variant (round 5): the default set's h_s knows nothing about y, so sigma / mu miss |y - mu| by a wide margin -
# 37 % of the y symbols leave their CDF row through the escape path and a 268 frame codes to 4.5 MB, far outside the
# regime of a trained model (~1 MB, escapes a rarity).  Without training, the hyper-decoder can still be made to emit
# sigma ~ rms(y) and mu ~ 0: its last LayerNorm gets a small gain and a fixed bias vector b, and the un-embed rows of
# the sigma channels are MATCHED_SIGMA x b / |b|^2 plus a small token-dependent part (so that sigma = MATCHED_SIGMA
# +- ~25 %: a spread of CDF-table rows like a trained model's), the mu rows a small random map.  Every other tensor is the
# default set's: the transformer work per frame is identical, only the entropy side changes.
MATCHED_SIGMA = 2.5          # rms(y) of the default set on N(0, 1) frames (bench.py `precision_f16.y_rms`: 2.49)
_MATCHED_KEYS = ("h_s.norm.weight", "h_s.norm.bias", "h_s.final.weight")


def _matched_tensor(key, shape, seed):
    rng = _rng(seed, "matched/" + key)
    if key == "h_s.norm.weight":
        return 0.25 * (1.0 + _randn(rng, shape, 0.1))
    if key == "h_s.norm.bias":
        return _randn(rng, shape, 1.0)
    # h_s.final.weight [F = zh*zw*2L, hd], rows ordered (p1 p2 c) (vit_nlc.py:665-680), c < L: sigma, c >= L: mu
    F, hd = shape
    b = _matched_tensor("h_s.norm.bias", (hd,), seed)
    w = _randn(rng, shape, 1.0 / np.sqrt(hd))
    w = w - torch.outer(w @ b, b) / float(b @ b)      # orthogonal to b: row . (gamma x^ + b) = constant + token part
    cout = F // 16 if F % 16 == 0 else F          # zh * zw = 16 in both model sizes
    c = torch.arange(F) % cout
    is_sigma = (c < cout // 2).to(torch.float32).unsqueeze(1)
    return is_sigma * (MATCHED_SIGMA * b / float(b @ b) + 2.4 * w) + (1.0 - is_sigma) * 0.4 * w


def synth_tensor(key, shape, seed=0, variant="default"):
    """Value of parameter `key` (reference naming) with `shape`; None = leave as is."""
    if key.endswith(_SKIP):
        return None
    if variant == "matched" and key in _MATCHED_KEYS:
        return _matched_tensor(key, tuple(shape), seed)
    if variant not in ("default", "matched"):
        raise ValueError(f"unknown synthetic weight variant {variant!r}")
    rng = _rng(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if key.startswith("entropy_bottleneck."):
        C = shape[0]
        if leaf.startswith("_matrix"):
            i = int(leaf[-1])
            filters = (1, 3, 3, 3, 3, 1)
            scale = 10.0 ** (1 / 5)
            init = float(np.log(np.expm1(1 / scale / filters[i + 1])))
            return init + _randn(rng, shape, 0.1)
        if leaf.startswith("_bias"):
            return torch.from_numpy(rng.uniform(-0.5, 0.5, size=shape).astype(np.float32))
        if leaf.startswith("_factor"):
            return _randn(rng, shape, 0.1)
        if leaf == "quantiles":
            med = rng.standard_normal(C).astype(np.float32) * 0.3
            lo = med - 4.0 * (1 + 0.3 * rng.random(C).astype(np.float32))
            hi = med + 4.0 * (1 + 0.3 * rng.random(C).astype(np.float32))
            return torch.from_numpy(np.stack([lo, med, hi], -1).reshape(C, 1, 3))
        return None
    if leaf in ("beta", "gamma") and (".beta_reparam" not in key and ".gamma_reparam" not in key):
        # GDN parameters (layers/gdn.py:41-74) in their re-parametrised form sqrt(v + pedestal)
        ped = np.float32(2.0 ** -36)
        if leaf == "beta":
            v = 1.0 + 0.5 * rng.random(size=shape, dtype=np.float32)
        else:
            v = 0.1 * np.eye(shape[0], dtype=np.float32) + 0.02 * np.abs(rng.standard_normal(size=shape, dtype=np.float32))
        return torch.from_numpy(np.sqrt(np.maximum(v + ped, ped)).astype(np.float32))
    if leaf == "pos_embed":
        return _randn(rng, shape, 0.2)
    if ".norm" in key and leaf == "weight" and len(shape) == 1:
        return 1.0 + _randn(rng, shape, 0.1)
    if leaf == "bias":
        return _randn(rng, shape, 0.05)
    if leaf == "weight":
        if key == "g_s.final.weight":  # ConvTranspose2d (in, out, kh, kw): fan-in = in
            return _randn(rng, shape, 1.0 / np.sqrt(shape[0]))
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if key.endswith(("attn.proj.weight", "mlp.fc2.weight")):
            gain = 0.5
        if key == "quant_conv.weight":
            gain = 1.5
        if key == "h_a.quan_mlp.fc2.weight":
            gain = 6.0
        if key == "h_s.final.weight":
            gain = 2.0
        if key in ("g_a.6.weight", "h_a.4.weight"):   # CNN zoo: last analysis convs (latents of O(few))
            gain = 4.0
        return _randn(rng, shape, gain / np.sqrt(fan_in))
    return None


@torch.no_grad()
def fill_state_dict(shapes, seed=0, device="cpu", variant="default"):
    """shapes: {key: shape}. Returns {key: tensor} for every synthesised key."""
    out = {}
    for k, shp in shapes.items():
        t = synth_tensor(k, shp, seed, variant)
        if t is not None:
            out[k] = t.to(torch.float32).to(device)
    return out


@torch.no_grad()
def apply_variant(net, seed=0, variant="matched"):
    """Switch a net that carries the synthetic weights of `seed` between the "default" and the "matched" entropy variant:
    only the three hyper-decoder tensors that differ are rewritten (the CDF tables do not depend on them)."""
    own = net.state_dict()
    for k in _MATCHED_KEYS:
        own[k].copy_(synth_tensor(k, tuple(own[k].shape), seed, variant).to(own[k].device))
    return net


def synth_frame(channels, seed, H=721, W=1440, kind="normal"):
    """A synthetic normalised ERA5 snapshot (C, 721, 1440) fp32: N(0,1) like
    normalised ERA5, or uniform [0,1) (README proxy input)."""
    g = torch.Generator().manual_seed(int(seed))
    if kind == "uniform":
        return torch.rand((channels, H, W), generator=g, dtype=torch.float32)
    return torch.randn((channels, H, W), generator=g, dtype=torch.float32)


def thin_model_kwargs():
    """Constructor arguments (reference signature, vaeformer.py:78-92) of the thin-width,
    full-spatial test model: 8 variables, width 128 (2 heads x 64), latent 16, hyper-prior
    width 144 (2 heads x 72) - same head dims, window pattern and 721x1440 geometry as the
    268 model at ~1/1000 of the FLOPs."""
    dd_kw = dict(z_dim=None, learnable_pos=True, window=True, window_size=[(24, 24), (12, 48), (48, 12)], interval=4,
                 drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                 test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=True, img_size=(721, 1440),
                 embed_dim=128, depth=8, num_heads=2)
    prior_kw = dict(z_dim=16, embed_dim=144, depth=4, num_heads=2, interval=1, learnable_pos=True, window=False,
                    drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                    test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=False, img_size=(72, 144))
    return dict(embed_dim=16, z_channels=16, y_channels=128, sample_posterior=False, frozen_encoder=False,
                lower_dim=True,
                ddconfig=dict(arch='vit_base', patch_size=(11, 10), patch_stride=(10, 10), in_chans=8, out_chans=8,
                              pretrained_model='', kwargs=dd_kw),
                priorconfig=dict(patch_size=(4, 4), in_chans=16, out_chans=16, pretrained_model='', kwargs=prior_kw))


@torch.no_grad()
def load_synthetic(net, seed=0, update=True, variant="default"):
    """Fill `net` (a VAEformer) with the deterministic synthetic weights and build its CDF
    tables (`update(force=True)`).  variant "matched": the entropy-representative set (see MATCHED_SIGMA)."""
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, seed, variant=variant)
    own = net.state_dict()
    for k, v in sd.items():
        own[k].copy_(v)
    if hasattr(net, "_derived"):
        net._derived.clear()
    if update:
        net.update(force=True)
    return net
