"""TEST INFRASTRUCTURE ONLY (oracle) - never imported by the product path.

CPU restatement (plain PyTorch-CPU ops, functional, state-dict driven) of the
reference's VAEformer encode/decode path.  Every function cites the reference
lines it follows (paths relative to /root/reference).  It is pinned against the
reference's own Python - imported in the build container through
oracle/ref_shim.py - by the fixtures under tests/golden/ (generator:
tests/golden/make_golden.py; check: tests/test_oracle_vs_golden.py).

It is also the `cpu_baseline` leg of bench.py (kind="port"): like the reference it
keeps the *math-path* attention with materialised scores
(cra5/models/vaeformer/vit_nlc.py:99-103).

The model is described by a plain dict `cfg` (see `cfg_268()` / `cfg_thin()`) and a
state dict `sd` with the reference's key names (vaeformer.py:148-159).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------


def cfg_268(in_chans=268):
    """cra5/models/vaeformer/vaeformer.py:93-142 (model_version == 268); the
    159-variable variant only changes in_chans/out_chans
    (config/vaeformer_era5_159v_1h.py)."""
    return dict(
        in_chans=in_chans, img_size=(721, 1440), patch_size=(11, 10), patch_stride=(10, 10),
        embed_dim=1024, depth=24, num_heads=16, window_size=[(24, 24), (12, 48), (48, 12)],
        interval=4, latent_dim=256,
        h_patch=(4, 4), h_embed_dim=360, h_depth=8, h_num_heads=5, z_dim=256,
    )


def cfg_thin():
    """Thin-width, full-spatial test model (same head dims 64 / 72 as the 268 model)."""
    return dict(
        in_chans=8, img_size=(721, 1440), patch_size=(11, 10), patch_stride=(10, 10),
        embed_dim=128, depth=8, num_heads=2, window_size=[(24, 24), (12, 48), (48, 12)],
        interval=4, latent_dim=16,
        h_patch=(4, 4), h_embed_dim=144, h_depth=4, h_num_heads=2, z_dim=16,
    )


def block_windows(first, last, interval, window_size):
    """vit_nlc.py:401-411 (encoder, i in [0, depth/2)) and :613-623 (decoder,
    i in [depth/2, depth)): block i is windowed iff (i+1) % interval != 0, with
    window_size[min(i % interval, len-1)]; otherwise global (None)."""
    out = []
    for i in range(first, last):
        if (i + 1) % interval != 0:
            out.append(tuple(window_size[min(i % interval, len(window_size) - 1)]))
        else:
            out.append(None)
    return out


def encoder_windows(cfg):
    w = block_windows(0, cfg["depth"] // 2, cfg["interval"], cfg["window_size"])
    return w + [w[-1]]  # vit_nlc.py:413-422: the last block is duplicated


def decoder_windows(cfg):
    return block_windows(cfg["depth"] // 2, cfg["depth"], cfg["interval"], cfg["window_size"])


# ----------------------------------------------------------------------------
# transformer pieces
# ----------------------------------------------------------------------------


def _ln(x, sd, pre):
    # partial(nn.LayerNorm, eps=1e-6): vit_nlc.py:381, 603
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], 1e-6)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def mlp(x, sd, pre):
    """vit_nlc.py:52-69 (nn.GELU() = exact erf form)."""
    return _lin(F.gelu(_lin(x, sd, pre + ".fc1")), sd, pre + ".fc2")


def _sdpa_math(q, k, v, scale):
    """vit_nlc.py:101-103 / :244-246: ((q*scale) @ k^T).softmax(-1) @ v."""
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    return attn @ v


def attention_global(x, sd, pre, heads):
    """vit_nlc.py:94-112, ATTENTION_MODE == 'math'."""
    B, N, C = x.shape
    hd = C // heads
    qkv = _lin(x, sd, pre + ".qkv").reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    o = _sdpa_math(qkv[0], qkv[1], qkv[2], hd ** -0.5)
    o = o.transpose(1, 2).reshape(B, N, C)
    return _lin(o, sd, pre + ".proj")


def attention_window(x, sd, pre, heads, H, W, ws):
    """vit_nlc.py:219-258: zero-pad bottom/right AFTER norm1 and BEFORE qkv (padded
    tokens carry q=k=v=bias), unmasked softmax per window, crop."""
    B, N, C = x.shape
    hd = C // heads
    wh, ww = ws
    x = x.reshape(B, H, W, C)
    pad_r = (ww - W % ww) % ww
    pad_b = (wh - H % wh) % wh
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = H + pad_b, W + pad_r
    # window_partition, vit_nlc.py:115-126
    xw = x.view(B, Hp // wh, wh, Wp // ww, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, wh * ww, C)
    Bw, Nw, _ = xw.shape
    qkv = _lin(xw, sd, pre + ".qkv").reshape(Bw, Nw, 3, heads, hd).permute(2, 0, 3, 1, 4)
    o = _sdpa_math(qkv[0], qkv[1], qkv[2], hd ** -0.5)
    o = o.transpose(1, 2).reshape(Bw, Nw, C)
    o = _lin(o, sd, pre + ".proj")
    # window_reverse, vit_nlc.py:129-142 (the view at :250 is a pure reinterpretation)
    o = o.view(B, Hp // wh, Wp // ww, wh, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    o = o[:, :H, :W, :].reshape(B, H * W, C)
    return o


def block(x, sd, pre, heads, H, W, ws):
    """vit_nlc.py:282-287."""
    h = _ln(x, sd, pre + ".norm1")
    if ws is None:
        a = attention_global(h, sd, pre + ".attn", heads)
    else:
        a = attention_window(h, sd, pre + ".attn", heads, H, W, ws)
    x = x + a
    x = x + mlp(_ln(x, sd, pre + ".norm2"), sd, pre + ".mlp")
    return x


# ----------------------------------------------------------------------------
# g_a / g_s / h_a / h_s
# ----------------------------------------------------------------------------


def g_a(x, sd, cfg):
    """ViT_Encoder.forward, vit_nlc.py:458-486 (z_dim None -> no quan_mlp).
    x: (B, C, 721, 1440) -> moments tokens (B, N, 2*D) [mean | logvar]."""
    t = F.conv2d(x, sd["g_a.patch_embed.proj.weight"], sd["g_a.patch_embed.proj.bias"],
                 stride=cfg["patch_stride"])
    Hp, Wp = t.shape[2], t.shape[3]
    t = t.flatten(2).transpose(1, 2) + sd["g_a.pos_embed"]
    wins = encoder_windows(cfg)
    nb = len(wins)
    for i in range(nb - 2):
        t = block(t, sd, f"g_a.blocks.{i}", cfg["num_heads"], Hp, Wp, wins[i])
    mean = block(t, sd, f"g_a.blocks.{nb - 2}", cfg["num_heads"], Hp, Wp, wins[nb - 2])
    logvar = block(t, sd, f"g_a.blocks.{nb - 1}", cfg["num_heads"], Hp, Wp, wins[nb - 1])
    return torch.cat([mean, logvar], 2), (Hp, Wp)


def encode_y(x, sd, cfg):
    """vaeformer.py:272-282: y = quant_conv(g_a(x))[:, :latent_dim] (posterior.mode(),
    modules/distributions.py:24-33,66-67). Returns y as NCHW."""
    mom, (Hp, Wp) = g_a(x, sd, cfg)
    B = mom.shape[0]
    w = sd["quant_conv.weight"][:, :, 0, 0]
    m = F.linear(mom, w, sd["quant_conv.bias"])  # 1x1 conv == per-token linear
    L = cfg["latent_dim"]
    y = m[:, :, :L].transpose(1, 2).reshape(B, L, Hp, Wp)
    return y


def h_a(y, sd, cfg):
    """HyperpriorEncoder, vit_nlc.py:488-551 (+ base forward :477-486)."""
    t = F.conv2d(y, sd["h_a.patch_embed.proj.weight"], sd["h_a.patch_embed.proj.bias"],
                 stride=cfg["h_patch"])
    Hz, Wz = t.shape[2], t.shape[3]
    t = t.flatten(2).transpose(1, 2) + sd["h_a.pos_embed"]
    for i in range(cfg["h_depth"] // 2):
        t = block(t, sd, f"h_a.blocks.{i}", cfg["h_num_heads"], Hz, Wz, None)
    t = mlp(t, sd, "h_a.quan_mlp")
    B, N, C = t.shape
    return t.reshape(B, Hz, Wz, C).permute(0, 3, 1, 2).contiguous()


def h_s(z_hat, sd, cfg):
    """HyperpriorDecoder, vit_nlc.py:696-748 with ViT_Decoder.forward :682-693 and
    up_forward :665-680 (Linear + 'b h w (p1 p2 c) -> b c (h p1) (w p2)').
    Returns gaussian params NCHW (B, 2*latent, 72, 144): scales first, means last
    (vaeformer.py:369)."""
    B, C, Hz, Wz = z_hat.shape
    t = z_hat.reshape(B, C, -1).permute(0, 2, 1)
    t = mlp(t, sd, "h_s.post_quan_mlp")
    for i in range(cfg["h_depth"] - cfg["h_depth"] // 2):
        t = block(t, sd, f"h_s.blocks.{i}", cfg["h_num_heads"], Hz, Wz, None)
    t = _ln(t, sd, "h_s.norm")
    t = F.linear(t, sd["h_s.final.weight"])
    p1, p2 = cfg["h_patch"]
    c_out = t.shape[-1] // (p1 * p2)
    t = t.reshape(B, Hz, Wz, p1, p2, c_out).permute(0, 5, 1, 3, 2, 4).reshape(B, c_out, Hz * p1, Wz * p2)
    return t


def g_s(y_hat, sd, cfg):
    """decode_latent, vaeformer.py:294-300: post_quant_conv -> ViT_Decoder
    (vit_nlc.py:655-693; no pos_embed) -> LayerNorm -> ConvTranspose2d (no bias)."""
    B, L, Hp, Wp = y_hat.shape
    t = y_hat.reshape(B, L, -1).permute(0, 2, 1)
    t = F.linear(t, sd["post_quant_conv.weight"][:, :, 0, 0], sd["post_quant_conv.bias"])
    wins = decoder_windows(cfg)
    for j, ws in enumerate(wins):
        t = block(t, sd, f"g_s.blocks.{j}", cfg["num_heads"], Hp, Wp, ws)
    t = _ln(t, sd, "g_s.norm")
    D = t.shape[-1]
    t = t.reshape(B, Hp, Wp, D).permute(0, 3, 1, 2)
    return F.conv_transpose2d(t, sd["g_s.final.weight"], None, stride=cfg["patch_stride"])


# ----------------------------------------------------------------------------
# entropy models
# ----------------------------------------------------------------------------

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64  # models/base.py:54-56


def get_scale_table():
    """models/base.py:59-61."""
    return torch.exp(torch.linspace(math.log(SCALES_MIN), math.log(SCALES_MAX), SCALES_LEVELS))


def eb_logits_cumulative(v, sd, pre="entropy_bottleneck"):
    """entropy_models.py:434-453. v: (C, 1, n)."""
    logits = v
    for i in range(5):
        logits = torch.matmul(F.softplus(sd[f"{pre}._matrix{i}"]), logits)
        logits = logits + sd[f"{pre}._bias{i}"]
        if i < 4:
            f = sd[f"{pre}._factor{i}"]
            logits = logits + torch.tanh(f) * torch.tanh(logits)
    return logits


def eb_likelihood(v, sd, pre="entropy_bottleneck"):
    """entropy_models.py:455-463 (this vendored copy: plain sigmoid difference)."""
    lower = eb_logits_cumulative(v - 0.5, sd, pre)
    upper = eb_logits_cumulative(v + 0.5, sd, pre)
    return torch.sigmoid(upper) - torch.sigmoid(lower), lower, upper


def eb_medians(sd, pre="entropy_bottleneck"):
    return sd[f"{pre}.quantiles"][:, 0, 1]  # entropy_models.py:390-392


def eb_forward(z, sd, pre="entropy_bottleneck"):
    """entropy_models.py:465-510, eval mode: z_hat = round(z - m) + m, likelihood
    lower-bounded at 1e-9."""
    B, C = z.shape[:2]
    m = eb_medians(sd, pre).reshape(1, C, *([1] * (z.dim() - 2)))
    z_hat = torch.round(z - m) + m
    v = z_hat.transpose(0, 1).reshape(C, 1, -1)
    lik, _, _ = eb_likelihood(v, sd, pre)
    lik = torch.clamp(lik, min=1e-9)
    lik = lik.reshape(C, B, *z.shape[2:]).transpose(0, 1)
    return z_hat, lik


def eb_symbols(z, sd, pre="entropy_bottleneck"):
    """entropy_models.py:529-535 + quantize(...,'symbols') :167-184."""
    C = z.shape[1]
    m = eb_medians(sd, pre).reshape(1, C, *([1] * (z.dim() - 2)))
    return torch.round(z - m).int()


def eb_tables(sd, pmf_to_cdf, pre="entropy_bottleneck"):
    """EntropyBottleneck.update, entropy_models.py:394-427 (+ _pmf_to_cdf :208-216).
    Returns (quantized_cdf int32 [C, L+2], cdf_length int32 [C], offset int32 [C])."""
    q = sd[f"{pre}.quantiles"]
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    offset = -minima
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max().item())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
    pmf, lower, upper = eb_likelihood(samples, sd, pre)
    pmf = pmf[:, 0, :]
    tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = _pmf_rows_to_cdf(pmf, tail_mass, pmf_length, max_length, pmf_to_cdf)
    return cdf, (pmf_length + 2).int(), offset.int()


def _pmf_rows_to_cdf(pmf, tail_mass, pmf_length, max_length, pmf_to_cdf):
    """entropy_models.py:208-216."""
    cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
    for i in range(len(pmf_length)):
        prob = torch.cat((pmf[i, : int(pmf_length[i])], tail_mass[i]), dim=0)
        c = pmf_to_cdf(prob.float().contiguous().numpy(), 16)
        cdf[i, : len(c)] = torch.from_numpy(np.asarray(c, dtype=np.int64)).int()
    return cdf


def _phi(u):
    """entropy_models.py:598-602: 0.5 * erfc(-(2**-0.5) * u)."""
    return 0.5 * torch.erfc(float(-(2 ** -0.5)) * u)


def gc_tables(scale_table, pmf_to_cdf, tail_mass=1e-9):
    """GaussianConditional.update, entropy_models.py:619-643."""
    import scipy.stats
    multiplier = -scipy.stats.norm.ppf(tail_mass / 2)
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(torch.max(pmf_length).item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    s = scale_table.unsqueeze(1).float()
    upper = _phi((0.5 - samples) / s)
    lower = _phi((-0.5 - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _pmf_rows_to_cdf(pmf, tail, pmf_length, max_length, pmf_to_cdf)
    return cdf, (pmf_length + 2).int(), (-pmf_center).int()


def gc_build_indexes(scales, scale_table, bound=0.11):
    """entropy_models.py:679-685 with LowerBound = max(x, bound) (ops/bound_ops.py:36-38)."""
    s = torch.max(scales, torch.tensor([bound], dtype=scales.dtype))
    idx = torch.full(s.shape, len(scale_table) - 1, dtype=torch.int32)
    for t in scale_table[:-1]:
        idx -= (s <= t).int()
    return idx


def gc_forward(y, scales, means, bound=0.11):
    """entropy_models.py:645-677, eval mode."""
    y_hat = torch.round(y - means) + means
    v = torch.abs(y_hat - means)
    s = torch.max(scales, torch.tensor([bound], dtype=scales.dtype))
    lik = _phi((0.5 - v) / s) - _phi((-0.5 - v) / s)
    return y_hat, torch.clamp(lik, min=1e-9)


def gc_symbols(y, means):
    """quantize(y, 'symbols', means), entropy_models.py:167-184."""
    return torch.round(y - means).int()


# ----------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------


def tables(sd, pmf_to_cdf):
    """CompressionModel.update(force=True), models/base.py:91-115."""
    st = get_scale_table()
    eb = eb_tables(sd, pmf_to_cdf)
    gc = gc_tables(st, pmf_to_cdf)
    return dict(eb_cdf=eb[0], eb_len=eb[1], eb_off=eb[2], gc_cdf=gc[0], gc_len=gc[1], gc_off=gc[2],
                scale_table=st)


def latent_side(y, sd, cfg, scale_table):
    """The part of compress()/compress_from_latent() between y and the coder
    (vaeformer.py:334-348): returns dict(z_sym, z_hat, scales, means, idx, y_sym, y_hat)."""
    z = h_a(y, sd, cfg)
    z_sym = eb_symbols(z, sd)
    C = z.shape[1]
    z_hat = z_sym.float() + eb_medians(sd).reshape(1, C, 1, 1)  # dequantize, :192-201
    params = h_s(z_hat, sd, cfg)
    scales, means = params.chunk(2, 1)
    idx = gc_build_indexes(scales, scale_table)
    y_sym = gc_symbols(y, means)
    y_hat = y_sym.float() + means
    return dict(z=z, z_sym=z_sym, z_hat=z_hat, scales=scales, means=means, idx=idx, y_sym=y_sym,
                y_hat=y_hat)


def compress(x, sd, cfg, tb, rans_encode):
    """VAEformer.compress, vaeformer.py:350-376, batch item 0 only (like the .bin
    writer, api/cra5_api.py:108-117). rans_encode(symbols, indexes, cdf, cdf_len, off)->bytes."""
    y = encode_y(x, sd, cfg)
    return compress_from_latent(y, sd, cfg, tb, rans_encode), y


def compress_from_latent(y, sd, cfg, tb, rans_encode):
    s = latent_side(y, sd, cfg, tb["scale_table"])
    C = s["z_sym"].shape[1]
    z_idx = torch.arange(C, dtype=torch.int32).reshape(1, C, 1, 1).expand_as(s["z_sym"])  # :512-523
    z_str = rans_encode(s["z_sym"][0].reshape(-1), z_idx[0].reshape(-1), tb["eb_cdf"], tb["eb_len"], tb["eb_off"])
    y_str = rans_encode(s["y_sym"][0].reshape(-1), s["idx"][0].reshape(-1), tb["gc_cdf"], tb["gc_len"], tb["gc_off"])
    return {"strings": [[y_str], [z_str]], "z_shape": tuple(s["z"].shape[-2:])}


def decompress(strings, z_shape, sd, cfg, tb, rans_decode, return_format="reconstructed"):
    """VAEformer.decompress, vaeformer.py:378-400."""
    C = sd["entropy_bottleneck.quantiles"].shape[0]
    z_idx = torch.arange(C, dtype=torch.int32).reshape(1, C, 1, 1).expand(1, C, *z_shape)
    z_sym = rans_decode(strings[1][0], z_idx.reshape(-1), tb["eb_cdf"], tb["eb_len"], tb["eb_off"])
    z_hat = z_sym.reshape(1, C, *z_shape).float() + eb_medians(sd).reshape(1, C, 1, 1)
    params = h_s(z_hat, sd, cfg)
    scales, means = params.chunk(2, 1)
    idx = gc_build_indexes(scales, tb["scale_table"])
    y_sym = rans_decode(strings[0][0], idx.reshape(-1), tb["gc_cdf"], tb["gc_len"], tb["gc_off"])
    y_hat = y_sym.reshape(means.shape).float() + means
    if return_format == "latent":
        return y_hat
    return {"x_hat": g_s(y_hat, sd, cfg)}


# ----------------------------------------------------------------------------
# GDN (not executed by VAEformer; north_star names it)
# ----------------------------------------------------------------------------


def gdn(x, beta_param, gamma_param, inverse=False, beta_min=1e-6, reparam_offset=2 ** -18):
    """layers/gdn.py:76-92 with NonNegativeParametrizer (ops/parametrizers.py:38-64)."""
    ped = reparam_offset ** 2
    beta = torch.clamp(beta_param, min=(beta_min + ped) ** 0.5) ** 2 - ped
    gamma = torch.clamp(gamma_param, min=(0 + ped) ** 0.5) ** 2 - ped
    C = x.shape[1]
    norm = F.conv2d(x ** 2, gamma.reshape(C, C, 1, 1), beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm
