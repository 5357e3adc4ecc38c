"""VAEformer codec model on MI355X: same object surface and state-dict layout as the
reference's `cra5.models.vaeformer.vaeformer.VAEformer` (vaeformer.py:70-403 in
taohan10200/CRA5), every tensor op executed by the hand-written HIP kernels of
`libcra5_amd.so` through the C ABI.  torch is used for parameter storage, device
buffers and streams only.

Reference behaviour kept on purpose:
  * block pattern W(24,24), W(12,48), W(48,12), G repeated, the duplicated last encoder
    block producing `mean` / `logvar` from the same input (vit_nlc.py:401-422, 463-475);
  * (48,12) windows zero-pad the 72-row grid to 96 AFTER norm1 / BEFORE qkv, unmasked
    (vit_nlc.py:229-246) - implemented inside the attention kernel (padded tokens read the
    qkv bias), so no padded GEMM rows are ever computed;
  * `y` = first half of quant_conv's output (posterior.mode()); the logvar half is dead
    work in the reference (distributions.py:24-33) and is simply not computed here;
  * z is "decoded" on the encode side so both sides see the same z_hat
    (vaeformer.py:365-366): de-quantising the symbols is bit-identical to decoding the
    stream (lossless coder), so the encode side skips the redundant rANS decode.
There is no CPU path: on a non-GPU device every compute method raises.
"""
import contextlib
import math
import threading
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import ERR_RANGE, Cra5Error, StreamDesyncError
from .config import RuntimeConfig
from .entropy import EntropyBottleneck, GaussianConditional, get_scale_table

__all__ = ["VAEformer", "config_for", "block_windows"]


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------

_DD_KW_268 = dict(z_dim=None, learnable_pos=True, window=True, window_size=[(24, 24), (12, 48), (48, 12)], interval=4,
                  drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                  test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=True, img_size=(721, 1440))
_PRIOR_KW_268 = dict(z_dim=256, embed_dim=360, depth=8, num_heads=5, interval=1, learnable_pos=True, window=False,
                     drop_path_rate=0., round_padding=True, pad_attn_mask=True,
                     test_pos_mode='learnable_simple_interpolate', lms_checkpoint_train=False, img_size=(72, 144))
_ARCH = dict(vit_base=dict(embed_dim=768, depth=12, num_heads=12),
             vit_large=dict(embed_dim=1024, depth=24, num_heads=16),
             vit_huge=dict(embed_dim=2048, depth=24, num_heads=16))  # vit_nlc.py:1000-1024


def config_for(model_version, embed_dim=None, z_channels=None, y_channels=None, ddconfig=None, priorconfig=None):
    """Resolve the reference's constructor arguments (vaeformer.py:78-142) into one flat
    dict.  model_version 268 is the reference's hard-wired model; 159 is the same
    architecture with 159 variables (config/vaeformer_era5_159v_1h.py) - the reference's
    zoo raises for it, we provide it."""
    if model_version in (268, 159):
        embed_dim, z_channels, y_channels = 256, 256, 1024
        ddconfig = dict(arch='vit_large', patch_size=(11, 10), patch_stride=(10, 10), in_chans=model_version,
                        out_chans=model_version, kwargs=dict(_DD_KW_268))
        priorconfig = dict(patch_size=(4, 4), in_chans=256, out_chans=256, kwargs=dict(_PRIOR_KW_268))
    if ddconfig is None or priorconfig is None:
        raise ValueError("VAEformer: model_version must be 268/159 or ddconfig/priorconfig must be given")
    dd = dict(_ARCH[ddconfig.get('arch', 'vit_base')])
    ddk = dict(ddconfig.get('kwargs') or {})
    dd.update({k: ddk[k] for k in ('embed_dim', 'depth', 'num_heads') if k in ddk})
    pr = dict(embed_dim=768, depth=12, num_heads=12)  # vit_nlc.py:1075-1082
    prk = dict(priorconfig.get('kwargs') or {})
    pr.update({k: prk[k] for k in ('embed_dim', 'depth', 'num_heads') if k in prk})
    ps = tuple(ddconfig['patch_size'])
    cfg = dict(
        in_chans=ddconfig['in_chans'], out_chans=ddconfig.get('out_chans', ddconfig['in_chans']),
        img_size=tuple(ddk.get('img_size', (721, 1440))), patch_size=ps,
        patch_stride=tuple(ddconfig.get('patch_stride') or ps),
        embed_dim=dd['embed_dim'], depth=dd['depth'], num_heads=dd['num_heads'],
        window_size=[tuple(w) for w in ddk.get('window_size', [(24, 24), (12, 48), (48, 12)])],
        interval=ddk.get('interval', 4), window=ddk.get('window', True),
        latent_dim=embed_dim, y_channels=y_channels, z_channels=z_channels,
        h_patch=tuple(priorconfig['patch_size']), h_in_chans=priorconfig['in_chans'],
        h_embed_dim=pr['embed_dim'], h_depth=pr['depth'], h_num_heads=pr['num_heads'],
        z_dim=prk.get('z_dim'), h_img_size=tuple(prk.get('img_size', (72, 144))),
    )
    if cfg['y_channels'] != cfg['embed_dim']:
        raise ValueError("y_channels must equal the encoder width (quant_conv input is 2*y_channels)")
    return cfg


def block_windows(first, last, interval, window_size, window=True):
    """vit_nlc.py:401-411 / :613-623: block i is windowed iff (i+1) % interval != 0."""
    out = []
    for i in range(first, last):
        if window and (i + 1) % interval != 0:
            out.append(tuple(window_size[min(i % interval, len(window_size) - 1)]))
        else:
            out.append(None)
    return out


# --------------------------------------------------------------------------------------
# parameter containers (names = reference state-dict keys)
# --------------------------------------------------------------------------------------


def _trunc_normal(shape, std=0.02):
    t = torch.empty(shape)
    nn.init.trunc_normal_(t, std=std)
    return t


class _Linear(nn.Module):
    def __init__(self, fin, fout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(_trunc_normal((fout, fin)))
        self.bias = nn.Parameter(torch.zeros(fout)) if bias else None


class _LayerNorm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Conv(nn.Module):
    def __init__(self, wshape, bias_n=None):
        super().__init__()
        fan_in = int(np.prod(wshape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        self.weight = nn.Parameter(torch.empty(wshape).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(bias_n).uniform_(-bound, bound)) if bias_n else None


class _Mlp(nn.Module):
    def __init__(self, fin, hidden, fout):
        super().__init__()
        self.fc1 = _Linear(fin, hidden)
        self.fc2 = _Linear(hidden, fout)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.qkv = _Linear(d, 3 * d)
        self.proj = _Linear(d, d)


class _Block(nn.Module):
    def __init__(self, d, heads, window, layer_id):
        super().__init__()
        self.norm1 = _LayerNorm(d)
        self.attn = _Attn(d)
        self.norm2 = _LayerNorm(d)
        self.mlp = _Mlp(d, 4 * d, d)
        self.heads = heads
        self.window = window  # (wh, ww) or None = global
        with torch.no_grad():  # fix_init_weight, vit_nlc.py:438-444
            self.attn.proj.weight.div_(math.sqrt(2.0 * layer_id))
            self.mlp.fc2.weight.div_(math.sqrt(2.0 * layer_id))


class _PatchEmbed(nn.Module):
    def __init__(self, cin, d, k):
        super().__init__()
        self.proj = _Conv((d, cin, k[0], k[1]), d)


def _sincos_pos_embed(d, grid):
    """get_2d_sincos_pos_embed, vit_nlc.py:906-956 (w goes first)."""
    gh, gw = grid
    ys, xs = np.meshgrid(np.arange(gh, dtype=np.float32), np.arange(gw, dtype=np.float32), indexing="ij")

    def one(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float32) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(d // 2, xs), one(d // 2, ys)], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


class _Encoder(nn.Module):
    """State of ViT_Encoder / HyperpriorEncoder (vit_nlc.py:328-551)."""

    def __init__(self, cin, d, heads, patch, grid, windows, z_dim=None):
        super().__init__()
        self.pos_embed = nn.Parameter(_sincos_pos_embed(d, grid))
        self.patch_embed = _PatchEmbed(cin, d, patch)
        self.blocks = nn.ModuleList([_Block(d, heads, w, i + 1) for i, w in enumerate(windows)])
        if z_dim is not None:  # vit_nlc.py:543-546
            self.quan_mlp = _Mlp(d, int(np.sqrt(d // z_dim)) * z_dim, z_dim)


class _Decoder(nn.Module):
    """State of ViT_Decoder / HyperpriorDecoder (vit_nlc.py:553-748)."""

    def __init__(self, d, heads, windows, final, z_dim=None):
        super().__init__()
        if z_dim is not None:  # vit_nlc.py:608-611
            self.post_quan_mlp = _Mlp(z_dim, int(np.sqrt(d // z_dim)) * z_dim, d)
        self.blocks = nn.ModuleList([_Block(d, heads, w, i + 1) for i, w in enumerate(windows)])
        self.norm = _LayerNorm(d)
        self.final = final


class _Posterior:
    """The slice of DiagonalGaussianDistribution the path uses (distributions.py:24-67)."""

    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------


def _rup(x, m):
    return (x + m - 1) // m * m


class _SlotGate:
    """At most `n` frames inside a GPU phase at a time; a waiter with the smaller `prio` number goes first (FIFO among
    equals).  The encode-side phase (g_a: the longest chain of host + GPU work still ahead of the frame) outranks the
    decode-side one (g_s: nothing after it): in a K-frame job the last frames' g_a then never queues behind earlier
    frames' g_s, whose kernels fill the GPU while those last frames sit in their host rANS phases - a shorter drain."""

    def __init__(self, n):
        self.n = int(n)
        self._free = int(n)
        self._cv = threading.Condition()
        self._waiters = []      # (prio, ticket)
        self._ticket = 0

    def acquire(self, prio=0):
        with self._cv:
            self._ticket += 1
            me = (prio, self._ticket)
            self._waiters.append(me)
            while not (self._free > 0 and min(self._waiters) == me):
                self._cv.wait()
            self._waiters.remove(me)
            self._free -= 1
            if self._free > 0 and self._waiters:
                self._cv.notify_all()

    def release(self):
        with self._cv:
            self._free += 1
            self._cv.notify_all()


class VAEformer(nn.Module):
    def __init__(self, model_version, embed_dim=None, z_channels=None, y_channels=None, sample_posterior=None,
                 pretrained_vae=None, frozen_encoder=None, ddconfig=None, priorconfig=None,
                 rate_distortion_loss=None, kl_loss=None, ignore_keys=(), lower_dim=False, **kwargs):
        super().__init__()
        if sample_posterior:
            raise NotImplementedError("sample_posterior=True is a training feature (vaeformer.py:279-280)")
        cfg = config_for(model_version, embed_dim, z_channels, y_channels, ddconfig, priorconfig)
        self.cfg = cfg
        self.sample_posterior = False
        self.lower_dim = True
        self.frozen_encoder = bool(frozen_encoder)
        D, L = cfg['embed_dim'], cfg['latent_dim']
        H, W = cfg['img_size']
        kh, kw = cfg['patch_size']
        sh, sw = cfg['patch_stride']
        self.Hp, self.Wp = H // sh, W // sw  # vit_nlc.py:299, 598
        if (self.Hp - 1) * sh + kh != H or (self.Wp - 1) * sw + kw != W:
            raise ValueError("img_size must be tiled exactly by the (overlapping) patches")
        zh, zw = cfg['h_patch']
        self.Hz, self.Wz = self.Hp // zh, self.Wp // zw

        enc_w = block_windows(0, cfg['depth'] // 2, cfg['interval'], cfg['window_size'], cfg['window'])
        enc_w = enc_w + [enc_w[-1]]  # vit_nlc.py:413-422
        dec_w = block_windows(cfg['depth'] // 2, cfg['depth'], cfg['interval'], cfg['window_size'], cfg['window'])

        self.entropy_bottleneck = EntropyBottleneck(cfg['z_channels'])
        self.g_a = _Encoder(cfg['in_chans'], D, cfg['num_heads'], (kh, kw), (self.Hp, self.Wp), enc_w)
        self.g_s = _Decoder(D, cfg['num_heads'], dec_w, _Conv((D, cfg['out_chans'], kh, kw)))
        self.quant_conv = _Conv((2 * L, 2 * cfg['y_channels'], 1, 1), 2 * L)
        self.post_quant_conv = _Conv((cfg['y_channels'], L, 1, 1), cfg['y_channels'])
        hd, hh = cfg['h_embed_dim'], cfg['h_num_heads']
        n_h = cfg['h_depth'] // 2
        self.h_a = _Encoder(cfg['h_in_chans'], hd, hh, (zh, zw), (self.Hz, self.Wz), [None] * n_h, cfg['z_dim'])
        self.h_s = _Decoder(hd, hh, [None] * (cfg['h_depth'] - n_h),
                            _Linear(hd, 2 * cfg['h_in_chans'] * zh * zw, bias=False), cfg['z_dim'])
        self.gaussian_conditional = GaussianConditional(None)
        # GPU phases of concurrent frames: exclusive (one frame's kernels at a time) or shared
        # (streams overlap: other frames' blocks fill the tail / epilogue gaps of a kernel)
        # Runtime settings: ONE RuntimeConfig object (cra5_amd/config.py), from the caller (`runtime=`) or from the
        # environment - the only place of the package that reads CRA5_* settings.  The attributes below are its fields,
        # kept as plain attributes so that a caller / bench.py can change them between frames.
        rc = kwargs.pop("runtime", None)
        self.runtime = rc = rc if rc is not None else RuntimeConfig.from_env()
        self.gpu_exclusive = rc.gpu_exclusive
        self._attn_mode = rc.attn_engine   # "split" (f16-MFMA, fp32-accurate) | "f32"
        # whole-grid attention: balanced 12-wave passes + key-split leftover (csrc/attention_split_f16.hip, BAL); False keeps
        # the plain 27-work-groups-per-head launch (a bit-identical-on-full-tiles alternative: tests flip the attribute)
        self.attn_balanced = True
        self._gemm_mode = rc.gemm_engine
        # range guard (csrc/split.h): frames whose split-f16 activations left the f16 range are re-run on the exact-f32
        # engines; counted here (encode side, decode side); range_guard False turns the re-run into an error
        self.range_fallbacks = [0, 0]
        self.range_guard = rc.range_guard
        # "fp32" (default): 3-product split, fp32-accurate.  "f16": BASELINE.json configs[4] -
        # g_a / g_s projections and attention use plain f16 operands (1 MFMA per product, fp32
        # accumulate); the hyper-prior / GaussianConditional side stays fp32-accurate so that
        # encoder and decoder derive identical CDF indexes.  RMSE-gated in tests/test_model_gpu.py.
        # optional timeline of the GPU phases: a list receives (thread id, t_request, t_start, t_end)
        # per phase (tools/phase_timeline.py); None = off
        self.phase_log = None
        # optional log of the host (rANS) phases: a list receives ("enc" | "dec_z" | "dec_y", seconds) per frame; None = off
        self.host_log = None
        self.precision = rc.precision
        # reduced-precision mode: "plain" (round 5) - activations and weights of g_a / g_s travel as PLAIN f16 rows wherever
        # the consuming kernel takes them (full 128-byte lines per k-step: -13..-18 % per GEMM launch, bit-identical
        # results); "split" keeps the hi planes of split rows (rounds 1-4; a test flips the attribute)
        self.f16_layout = "plain"
        self._derived = {}
        self._derive_lock = threading.RLock()
        self._gpu_lock = threading.Lock()
        self._fallback_lock = threading.Lock()     # range-guard re-runs on the exact-f32 engines: one frame at a time
        # gpu_slots = n > 0: at most n frames inside a GPU phase at a time (shared-stream mode only):
        # keeps a kernel's tail filled by another frame's blocks without letting ALL frames fall
        # into the host (rANS) phase together, which idles the GPU.  0 = unlimited.
        self.gpu_slots = rc.gpu_slots
        # Bit-identical implementation alternatives, kept because tests compare the two forms (attributes, no environment
        # variable): y symbols resolved against the CDF tables by a device kernel (same byte stream) | on the host;
        # decode side: uint8 CDF indexes / int16 symbols between device and host coder | the int32 records of the
        # reference's interface; un-embed: GEMM epilogue scatters straight into the reconstruction
        # (csrc/gemm_split_epilogue_unembed.inc) | GEMM -> column matrix -> overlap-add
        self.resolve_on_gpu = True
        self.compact_records = True
        self.fused_unembed = True
        self._gpu_sem = None
        # (frames waiting for a GPU-phase slot: encode-side phases first - see _SlotGate; the ~1 ms h_s phase between the
        # two host phases of a decode does not queue for a slot and runs on a high-priority stream.  Rounds 3-5 carried
        # environment switches for the alternatives - decode-side first / arrival order, queueing h_s - which measured
        # slower every time: profiles/EXPERIMENTS.md)
        self._tls = threading.local()  # per-thread workspaces: one frame pipeline per thread/stream
        self.eval()


    # ---- engines: model-wide setting, overridable per thread (the range guard re-runs ONE frame on the exact-f32
    # engines while the other frame threads of the pipeline keep the split engines) ------------------------------
    @property
    def gemm_mode(self):
        o = getattr(self._tls, "engine_override", None)
        return o if o is not None else self._gemm_mode

    @gemm_mode.setter
    def gemm_mode(self, v):
        if v not in ("split", "f32"):
            raise ValueError("gemm_mode must be 'split' or 'f32'")
        self._gemm_mode = v

    @property
    def attn_mode(self):
        o = getattr(self._tls, "engine_override", None)
        return o if o is not None else self._attn_mode

    @attn_mode.setter
    def attn_mode(self, v):
        self._attn_mode = v

    @contextlib.contextmanager
    def _exact_f32_engines(self):
        prev = getattr(self._tls, "engine_override", None)
        self._tls.engine_override = "f32"
        try:
            yield
        finally:
            self._tls.engine_override = prev

    def _probe(self, *items, name="probe", out=None):
        """Device-side finiteness probe, asynchronous: ops.PROBE_PARTIALS partial sums per argument (a tensor, or (tensor,
        stride) for a strided sample) written by ONE product kernel each into this thread's probe buffer - a sum is
        non-finite as soon as one addend is (fp32 sums of O(1e7) bounded activations do not overflow).  The caller copies
        the values to the host with the phase's other results and tests them there (`_finite`).  (Rounds 1-4 used torch
        reductions + stack / cat here: the last torch compute kernels on the frame path.)"""
        P = ops.PROBE_PARTIALS
        buf = out if out is not None else self._buf(f"{name}{len(items)}", (len(items) * P,))
        for k, it in enumerate(items):
            t, stride = it if isinstance(it, tuple) else (it, 1)
            ops.probe_sums(t if t.is_contiguous() else t.contiguous(), buf[k * P:(k + 1) * P], stride)
        return buf

    @staticmethod
    def _hs_parent(scales, means):
        """mu and sigma are the two halves of ONE tensor written by h_s's un-embed: probe the parent (one reduction)."""
        b = means._base
        if b is not None and b is scales._base and b.is_contiguous() and b.numel() == means.numel() + scales.numel():
            return (b,)
        return (scales, means)

    @staticmethod
    def _finite(host_values):
        return bool(torch.isfinite(host_values).all())

    def _range_guard(self, side, run, what):
        """Run one frame's GPU work `run()` -> (result, finite flag [device bool], hyper flag or None).  A non-finite
        result means a split-f16 store left the f16 range (csrc/split.h poisons it) or the input itself was non-finite:
        the frame is re-run on the exact-f32 engines (fp32 operands have no such range).  Still non-finite, or the
        hyper-prior path (pinned engine: both sides of the codec must run the same kernels) non-finite: an error -
        never a quietly degraded frame."""
        res, ok, ok_h = run()
        if bool(ok) and (ok_h is None or bool(ok_h)):
            return res
        if ok_h is not None and bool(ok) and not bool(ok_h):
            raise FloatingPointError(f"{what}: the hyper-prior path produced non-finite entropy parameters (its engine is "
                                     "pinned on both sides of the codec, there is no fallback): check the checkpoint")
        if self.gemm_mode == "f32" or not self.range_guard:
            raise FloatingPointError(f"{what}: non-finite values on the device (" + (
                "exact-f32 engines: the input or the weights are non-finite" if self.gemm_mode == "f32" else
                "CRA5_RANGE_GUARD=0: a split-f16 activation left the f16 range, or the input is non-finite") + ")")
        import warnings
        with self._gpu_lock:                      # (12 frame threads share the counters)
            self.range_fallbacks[side] += 1
            n_fb = sum(self.range_fallbacks)
        # the count is part of the text: Python shows a repeated identical warning once per call site, and EVERY fallback
        # should be visible - a checkpoint that sends many frames here runs at the exact-f32 engines' speed (~1/4)
        warnings.warn(f"{what}: non-finite values with the split-f16 engines (an activation beyond 65 504, csrc/split.h) - "
                      f"re-running this frame on the exact-f32 engines (fallback #{n_fb} of this model)", RuntimeWarning,
                      stacklevel=3)
        # Fallbacks run ONE AT A TIME: the exact-f32 engines bring their own per-thread fp32 workspaces (qkv 127 MB, hidden
        # 170 MB, un-embed columns 315 MB for the 268 model) and, on first use, fp32 copies of every g_a / g_s weight
        # (1.6 GB, built once under _derive_lock) - bounded extra memory however many of the in-flight frames overflow.
        with self._fallback_lock, self._exact_f32_engines():
            res, ok, ok_h = run()
        if not (bool(ok) and (ok_h is None or bool(ok_h))):
            raise FloatingPointError(f"{what}: non-finite values with the exact-f32 engines too - the input frame (or the "
                                     "latent handed in) holds NaN / inf, or the checkpoint does")
        return res

    # ---- reference-compatible loading (vaeformer.py:168-185, base.py:69-89) -------------
    @classmethod
    def from_state_dict(cls, state_dict):
        variable_num = state_dict["backbone.g_a.patch_embed.proj.weight"].size(1)
        new_sd = OrderedDict()
        for k, v in state_dict.items():
            if 'kl_loss.logvar' not in k:
                new_sd[k.replace("backbone.", "")] = v
        net = cls(variable_num)
        net.load_state_dict(new_sd)
        return net

    def load_state_dict(self, state_dict, strict=True):
        """Resizes the CDF buffers to the checkpoint's (models/base.py:69-89)."""
        for name, names in (("entropy_bottleneck", ("_quantized_cdf", "_offset", "_cdf_length")),
                            ("gaussian_conditional", ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"))):
            mod = getattr(self, name)
            for b in names:
                key = f"{name}.{b}"
                if key in state_dict:
                    buf = getattr(mod, b)
                    if buf.numel() == 0:
                        buf.resize_(state_dict[key].size())
        self._derived.clear()
        return nn.Module.load_state_dict(self, state_dict, strict=strict)

    def update(self, scale_table=None, force=False):
        """CompressionModel.update (models/base.py:91-115)."""
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.entropy_bottleneck.update(force=force)
        updated |= self.gaussian_conditional.update_scale_table(scale_table, force=force)
        return updated

    @property
    def downsampling_factor(self):
        return 2 ** (4 + 2)

    # ---- device plumbing ---------------------------------------------------------------
    @property
    def device(self):
        d = self.__dict__.get("_dev")   # (cached: asked ~230 times per frame by the workspace lookups; _apply() resets it)
        if d is None:
            d = self.__dict__["_dev"] = self.quant_conv.weight.device
        return d

    def _apply(self, fn, *args, **kwargs):
        try:
            return super()._apply(fn, *args, **kwargs)
        finally:
            self.__dict__["_dev"] = None

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("cra5_amd.VAEformer computes only on an MI355X (HIP kernels, no CPU fallback). "
                               "Move the model with .to('cuda'); the CPU restatement lives in oracle/ (tests only).")

    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        """Persistent workspace, allocated once per (thread, device): concurrent frame
        pipelines (cra5_amd/pipeline.py: one host thread + one HIP stream per in-flight
        frame) never share activation buffers; weights are shared read-only."""
        ws = getattr(self._tls, "ws", None)
        if ws is None:
            ws = self._tls.ws = {}
        key = (name, tuple(shape), dtype, self.device)
        b = ws.get(name)
        if b is None or b[0] != key:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=dtype)
            ws[name] = (key, t)
            return t
        return b[1]

    def _derive(self, name, src, fn):
        """GEMM-ready re-layouts of weights, cached until the parameter changes."""
        key = (src.data_ptr(), src._version, src.device)
        d = self._derived.get(name)
        if d is None or d[0] != key:
            with self._derive_lock:
                d = self._derived.get(name)
                if d is None or d[0] != key:
                    with torch.no_grad():
                        d = (key, fn(src.detach()))
                    torch.cuda.current_stream().synchronize()  # visible to every stream
                    self._derived[name] = d
        return d[1]

    # ---- GEMM engine ------------------------------------------------------------------------
    # gemm_mode "split" (default): every projection runs on the f16 matrix cores with fp32
    # operands split into hi/lo halves (csrc/gemm_split_f16.hip: fp32-class accuracy at ~4x the
    # exact-f32 MFMA rate); the producers (LayerNorm, GELU epilogue, attention, patch gather)
    # emit the split layout directly.  gemm_mode "f32": the exact v_mfma_f32_32x32x2_f32 kernel
    # (csrc/gemm_f32.hip) - kept as the bit-for-bit-fmaf reference engine.
    def _sbuf(self, name, rows, K, zero=False):
        ws = getattr(self._tls, "sws", None)
        if ws is None:
            ws = self._tls.sws = {}
        key = (rows, K, self.device)
        b = ws.get(name)
        if b is None or b[0] != key:
            b = (key, ops.SplitMat.empty(rows, K, self.device, zero=zero))
            ws[name] = b
        return b[1]

    def _weight2d(self, key, w):
        """GEMM-ready [N, K] view / re-layout of a parameter (cached)."""
        if key == "g_s.final":     # ConvTranspose2d (D, C, kh, kw) -> W[N = C*kh*kw][K = D]
            return self._derive("w2d." + key, w, lambda w: w.reshape(w.shape[0], -1).t().contiguous())
        return w.detach().reshape(w.shape[0], -1)

    def _wsplit(self, key, w):
        """Split-f16 copy of a weight, scaled by a per-tensor power of two (cached)."""
        return self._derive("ws." + key, w, lambda w: ops.split_f16(self._weight2d(key, w).contiguous(), "auto"))

    def _wplain(self, key, w):
        """PLAIN f16 copy of a weight (the hi plane of its split copy, same power-of-two scale), cached."""
        return self._derive("wpl." + key, w, lambda _w: self._wsplit(key, w).plain_copy())

    def _plain_gemm(self, key, M, N, K):
        """Does the reduced-precision GEMM `key` of shape M x N x K run on plain-f16 operands?"""
        return (self.precision == "f16" and self.f16_layout == "plain" and self.gemm_mode == "split"
                and key.startswith(("g_a.", "g_s.")) and ops.plain_ok(M, N, _rup(K, 32)))

    def _wf32(self, key, w, pad32=False):
        w2 = self._weight2d(key, w)
        if pad32 and w2.shape[1] % 32:
            def mk(_):
                out = torch.zeros((w2.shape[0], _rup(w2.shape[1], 32)), device=w2.device, dtype=torch.float32)
                out[:, : w2.shape[1]] = w2
                return out
            return self._derive("wp." + key, w, mk)
        return w2

    def _act(self, t, name):
        """fp32 activation [rows, K] -> GEMM input handle for the current engine."""
        if self.gemm_mode == "split":
            return ops.split_f16(t, out=self._sbuf(name, t.shape[0], t.shape[1]))
        return t

    def _ln(self, x, norm, name, plain=False):
        rows, D = x.shape
        if self.gemm_mode == "split":
            sm = self._sbuf(name, rows, D)
            ops.layernorm(x, norm.weight, norm.bias, 1e-6, out_split=sm, want_f32=False, out_plain=plain)
            return sm
        return ops.layernorm(x, norm.weight, norm.bias, 1e-6, out=self._buf(name, (rows, D)))

    def _mm(self, a, key, w, bias=None, res=None, gelu=False, out=None, out_name=None, out_plain=False):
        """epi(a @ W^T).  `out`: fp32 destination (tensor / strided view) or None; `out_name`:
        produce the result as the next GEMM's input (split engine: written split by the
        epilogue, no fp32 copy; f32 engine: a named fp32 workspace).  out_plain: reduced-precision mode, the named
        result's rows are plain f16 (the caller has checked that its consumer takes them)."""
        if self.gemm_mode == "split":
            W = self._wsplit(key, w)
            hi = self.precision == "f16" and key.startswith(("g_a.", "g_s."))
            if self._plain_gemm(key, a.rows, W.rows, W.K):
                W = self._wplain(key, w)
            else:
                assert not a.plain and not out_plain, key
            if out_name is not None:
                sm = self._sbuf(out_name, a.rows, W.rows)
                ops.gemm_nt_split(a, W, bias=bias, res=res, gelu=gelu, out_split=sm, want_f32=False, hi_only=hi,
                                  out_plain=out_plain)
                return sm
            return ops.gemm_nt_split(a, W, bias=bias, res=res, gelu=gelu, out=out, hi_only=hi)
        W = self._wf32(key, w, pad32=(a.shape[1] % 32 == 0 and self._weight2d(key, w).shape[1] != a.shape[1]))
        if out_name is not None:
            out = self._buf(out_name, (a.shape[0], W.shape[0]))
        return ops.gemm_nt(a, W, bias=bias, res=res, gelu=gelu, out=out)

    # ---- transformer block on device -----------------------------------------------------
    def _block(self, blk, pre, t_in, t_out, grid):
        """vit_nlc.py:282-287. t_in: [N, D] fp32 residual stream; t_out: [N, D] (may be t_in, may
        be a strided view) receives x + attn(LN(x)) + mlp(LN(.))."""
        N, D = t_in.shape
        H, W = grid
        split = self.gemm_mode == "split"
        wh, ww = blk.window if blk.window is not None else (H, W)
        fused_attn = split and self.attn_mode == "split" and ops.split_attention_ok(D, blk.heads, wh, ww, H, W)
        # reduced-precision mode: which matrices of the block travel as plain f16 rows - a producer writes plain only
        # when its consumer's GEMM takes plain operands (and, for an epilogue, when its own launch is the wide form)
        p_qkv, p_proj = (self._plain_gemm(pre + ".attn.qkv", N, 3 * D, D), self._plain_gemm(pre + ".attn.proj", N, D, D))
        p_fc1, p_fc2 = (self._plain_gemm(pre + ".mlp.fc1", N, 4 * D, D), self._plain_gemm(pre + ".mlp.fc2", N, D, 4 * D))
        h = self._ln(t_in, blk.norm1, f"h{D}", plain=p_qkv)
        if fused_attn:
            # qkv never exists in fp32: GEMM epilogue -> split-f16 -> f16-MFMA attention -> split
            qkv_s = self._mm(h, pre + ".attn.qkv", blk.attn.qkv.weight, bias=blk.attn.qkv.bias, out_name=f"qkv{D}",
                             out_plain=p_qkv and p_proj)
            if qkv_s.plain:
                pad_s = self._derive("padp." + pre, blk.attn.qkv.bias, lambda b: ops.split_f16(b.reshape(1, -1)).plain_copy())
            else:
                pad_s = self._derive("pad." + pre, blk.attn.qkv.bias, lambda b: ops.split_f16(b.reshape(1, -1)))
            att = self._sbuf(f"att{D}", N, D, zero=True)
            ws, bal = None, False
            if blk.window is None and self.attn_balanced:
                # (g_a / g_s outputs depend on the CU count and this schedule by fp32 rounding - include/cra5_amd.h;
                # only the hyper-prior path, which never takes this branch, must be bit-stable across the codec's sides)
                bal, nb = ops.attention_balanced_plan(N, blk.heads)
                ws = self._buf("attn_ws", (nb,), torch.uint8) if (bal and nb) else None
            ops.window_attention_split(qkv_s, pad_s, blk.heads, H, W, wh, ww, out_split=att, workspace=ws, balanced=bal,
                                       hi_only=self.precision == "f16" and pre.startswith(("g_a.", "g_s.")))
            self._mm(att, pre + ".attn.proj", blk.attn.proj.weight, bias=blk.attn.proj.bias, res=t_in, out=t_out)
            h = self._ln(t_out, blk.norm2, f"h{D}", plain=p_fc1)
            hid = self._mm(h, pre + ".mlp.fc1", blk.mlp.fc1.weight, bias=blk.mlp.fc1.bias, gelu=True,
                           out_name=f"hid{D}", out_plain=p_fc1 and p_fc2)
            self._mm(hid, pre + ".mlp.fc2", blk.mlp.fc2.weight, bias=blk.mlp.fc2.bias, res=t_out, out=t_out)
            return t_out
        qkv = self._mm(h, pre + ".attn.qkv", blk.attn.qkv.weight, bias=blk.attn.qkv.bias,
                       out=self._buf(f"qkv{D}", (N, 3 * D)))
        if split:
            att = self._sbuf(f"att{D}", N, D, zero=True)   # pad columns stay zero
            ops.window_attention(qkv, blk.attn.qkv.bias, blk.heads, H, W, wh, ww, out_split=att, want_f32=False)
        else:
            att = ops.window_attention(qkv, blk.attn.qkv.bias, blk.heads, H, W, wh, ww,
                                       out=self._buf(f"att{D}", (N, D)))
        self._mm(att, pre + ".attn.proj", blk.attn.proj.weight, bias=blk.attn.proj.bias, res=t_in, out=t_out)
        h = self._ln(t_out, blk.norm2, f"h{D}")
        hid = self._mm(h, pre + ".mlp.fc1", blk.mlp.fc1.weight, bias=blk.mlp.fc1.bias, gelu=True, out_name=f"hid{D}")
        self._mm(hid, pre + ".mlp.fc2", blk.mlp.fc2.weight, bias=blk.mlp.fc2.bias, res=t_out, out=t_out)
        return t_out

    # ---- g_a + quant_conv -----------------------------------------------------------------
    def _encode_y_frame(self, x, mean=None, std=None):
        """x: [C, H, W] on device -> y [L, Hp, Wp] (vaeformer.py:272-282)."""
        cfg = self.cfg
        D, L = cfg['embed_dim'], cfg['latent_dim']
        N = self.Hp * self.Wp
        kh, kw = cfg['patch_size']
        sh, sw = cfg['patch_stride']
        K = cfg['in_chans'] * kh * kw
        x = x.contiguous()
        if self.gemm_mode == "split":
            cols = self._sbuf("pe_cols", N, K, zero=True)   # K padding written once, never touched
            # (reduced-precision mode: plain rows - half the bytes written, full lines for the patch-embed GEMM; the ERA5
            # geometry's tiled gather re-zeroes a plain row's padding itself)
            p_pe = (kh, kw, sh, sw) == (11, 10, 10, 10) and self.Wp % 16 == 0 and \
                self._plain_gemm("g_a.patch_embed.proj", N, D, K)
            ops.im2col(x, kh, kw, sh, sw, mean=mean, std=std, out_split=cols, out_plain=p_pe)
        else:
            cols = self._buf("pe_cols", (N, _rup(K, 32)), zero=True)
            ops.im2col(x, kh, kw, sh, sw, ldk=cols.shape[1], mean=mean, std=std, out=cols)
        t = self._buf(f"t{D}", (N, D))
        self._mm(cols, "g_a.patch_embed.proj", self.g_a.patch_embed.proj.weight, bias=self.g_a.patch_embed.proj.bias,
                 res=self.g_a.pos_embed[0], out=t)
        blocks = self.g_a.blocks
        grid = (self.Hp, self.Wp)
        nb = len(blocks)
        for i in range(nb - 2):
            self._block(blocks[i], f"g_a.blocks.{i}", t, t, grid)
        mom = self._buf("mom", (N, 2 * D))
        self._block(blocks[nb - 2], f"g_a.blocks.{nb - 2}", t, mom[:, :D], grid)   # mean
        self._block(blocks[nb - 1], f"g_a.blocks.{nb - 1}", t, mom[:, D:], grid)   # logvar (vit_nlc.py:468-470)
        # only the `mean` half of quant_conv's output is ever used (posterior.mode())
        wq = self._derive("wq_mean", self.quant_conv.weight, lambda w: w.view(2 * L, -1)[:L].contiguous())
        ytok = self._mm(self._act(mom, "mom_s"), "quant_conv.mean", wq, bias=self.quant_conv.bias[:L],
                        out=self._buf("ytok", (N, L)))
        y = torch.empty((L, self.Hp, self.Wp), device=self.device, dtype=torch.float32)
        ops.transpose(ytok, out=y.view(L, N))
        return y

    # ---- hyper-prior ----------------------------------------------------------------------
    # h_a / h_s always run on ONE fixed engine - the small-M split-f16 GEMM and the exact-fp32 attention
    # of csrc/hyper.hip - whatever CRA5_GEMM / CRA5_ATTN / CRA5_GEMM_TILE / the precision mode say: the
    # decoder re-derives the CDF indexes from h_s and a single flipped index desynchronises the rANS
    # stream, so the encode and the decode side must not be able to pick different kernels.
    def _hy_mm(self, a, key, w, bias=None, res=None, gelu=False, out=None, out_name=None, unembed=None):
        W = self._wsplit(key, w)
        if out_name is not None:
            sm = self._sbuf(out_name, a.rows, W.rows)
            ops.small_gemm_nt_split(a, W, bias=bias, res=res, gelu=gelu, out_split=sm, want_f32=False)
            return sm
        return ops.small_gemm_nt_split(a, W, bias=bias, res=res, gelu=gelu, out=out, unembed=unembed)

    def _hy_ln(self, x, norm, name):
        sm = self._sbuf(name, x.shape[0], x.shape[1])
        ops.layernorm(x, norm.weight, norm.bias, 1e-6, out_split=sm, want_f32=False)
        return sm

    def _hy_block(self, blk, pre, t):
        """vit_nlc.py:282-287 with global attention (vit_nlc.py:94-112) on the [n, d] fp32 stream t, in place."""
        n, d = t.shape
        h = self._hy_ln(t, blk.norm1, f"hy_h{d}")
        qkv = self._hy_mm(h, pre + ".attn.qkv", blk.attn.qkv.weight, bias=blk.attn.qkv.bias,
                          out=self._buf(f"hy_qkv{d}", (n, 3 * d)))
        att = self._sbuf(f"hy_att{d}", n, d, zero=True)   # pad columns stay zero
        if (d // blk.heads) in (64, 72):
            ops.hyper_attention(qkv, blk.heads, out_split=att, want_f32=False)
        else:   # other head dims: the generic exact-fp32 kernel (same on both sides)
            ops.window_attention(qkv, blk.attn.qkv.bias, blk.heads, self.Hz, self.Wz, self.Hz, self.Wz, out_split=att,
                                 want_f32=False)
        self._hy_mm(att, pre + ".attn.proj", blk.attn.proj.weight, bias=blk.attn.proj.bias, res=t, out=t)
        h = self._hy_ln(t, blk.norm2, f"hy_h{d}")
        hid = self._hy_mm(h, pre + ".mlp.fc1", blk.mlp.fc1.weight, bias=blk.mlp.fc1.bias, gelu=True,
                          out_name=f"hy_hid{d}")
        self._hy_mm(hid, pre + ".mlp.fc2", blk.mlp.fc2.weight, bias=blk.mlp.fc2.bias, res=t, out=t)

    def _h_a_frame(self, y):
        """y [L, Hp, Wp] -> z [Cz, Hz*Wz] (vit_nlc.py:488-551)."""
        cfg = self.cfg
        d = cfg['h_embed_dim']
        zh, zw = cfg['h_patch']
        n = self.Hz * self.Wz
        K = y.shape[0] * zh * zw
        cols = ops.im2col(y, zh, zw, zh, zw, out_split=self._sbuf("hcols", n, K, zero=True))
        t = self._buf(f"hy_t{d}", (n, d))
        self._hy_mm(cols, "h_a.patch_embed.proj", self.h_a.patch_embed.proj.weight,
                    bias=self.h_a.patch_embed.proj.bias, res=self.h_a.pos_embed[0], out=t)
        for i, blk in enumerate(self.h_a.blocks):
            self._hy_block(blk, f"h_a.blocks.{i}", t)
        m = self.h_a.quan_mlp
        u = self._hy_mm(ops.split_f16(t, out=self._sbuf("ha_t_s", n, d)), "h_a.quan_mlp.fc1", m.fc1.weight,
                        bias=m.fc1.bias, gelu=True, out_name="ha_u")
        ztok = self._hy_mm(u, "h_a.quan_mlp.fc2", m.fc2.weight, bias=m.fc2.bias)
        return ops.transpose(ztok)  # [Cz, n]

    def _h_s_frame(self, z_hat):
        """z_hat [Cz, n] -> (scales, means), each [L, Hp, Wp] (vit_nlc.py:696-748, 665-680;
        chunk order vaeformer.py:369).  Deterministic: same kernels / reduction order on the
        encode and the decode side."""
        cfg = self.cfg
        d = cfg['h_embed_dim']
        zh, zw = cfg['h_patch']
        n = self.Hz * self.Wz
        ztok = ops.transpose(z_hat)  # [n, Cz]
        m = self.h_s.post_quan_mlp
        u = self._hy_mm(ops.split_f16(ztok, out=self._sbuf("hs_z_s", n, ztok.shape[1])), "h_s.post_quan_mlp.fc1",
                        m.fc1.weight, bias=m.fc1.bias, gelu=True, out_name="hs_u")
        t = self._buf(f"hy_t{d}", (n, d))
        self._hy_mm(u, "h_s.post_quan_mlp.fc2", m.fc2.weight, bias=m.fc2.bias, out=t)
        for i, blk in enumerate(self.h_s.blocks):
            self._hy_block(blk, f"h_s.blocks.{i}", t)
        h = self._hy_ln(t, self.h_s.norm, f"hy_h{d}")
        F = self.h_s.final.weight.shape[0]
        cout = F // (zh * zw)
        if zw == 4:
            # un-embed `b h w (p1 p2 c) -> b c (h p1) (w p2)` fused into the GEMM store: weight rows
            # re-ordered to (c, p1, p2) once, a lane writes 4 horizontal pixels of the image
            wps = self._derive("ws.h_s.final.ps", self.h_s.final.weight, lambda w: ops.split_f16(
                w.view(zh, zw, cout, -1).permute(2, 0, 1, 3).reshape(F, -1).contiguous(), "auto"))
            params = torch.empty((cout, self.Hz * zh, self.Wz * zw), device=self.device, dtype=torch.float32)
            ops.small_gemm_nt_split(h, wps, out=params, unembed=(self.Hz, self.Wz, zh, zw))
        else:
            lin = self._hy_mm(h, "h_s.final", self.h_s.final.weight)
            params = ops.pixel_shuffle(lin, self.Hz, self.Wz, zh, zw)  # [2L, Hp, Wp]
        L = params.shape[0] // 2
        return params[:L], params[L:]

    # ---- g_s --------------------------------------------------------------------------------
    def _decode_frame(self, y_hat, mean=None, std=None):
        """y_hat [L, Hp, Wp] -> x_hat [C, H, W] (vaeformer.py:294-300)."""
        cfg = self.cfg
        D, L = cfg['embed_dim'], cfg['latent_dim']
        N = self.Hp * self.Wp
        kh, kw = cfg['patch_size']
        sh, sw = cfg['patch_stride']
        ytok = self._buf("ytok", (N, L))
        ops.transpose(y_hat.reshape(L, N), out=ytok)
        t = self._buf(f"t{D}", (N, D))
        self._mm(self._act(ytok, "ytok_s"), "post_quant_conv", self.post_quant_conv.weight,
                 bias=self.post_quant_conv.bias, out=t)
        for j, blk in enumerate(self.g_s.blocks):
            self._block(blk, f"g_s.blocks.{j}", t, t, (self.Hp, self.Wp))
        Cout, (Himg, Wimg) = cfg['out_chans'], cfg['img_size']
        # reduced-precision mode: the fused un-embed (always the 256 x 256 wide form) takes plain operands when D % 64 == 0
        p_ue = (self.precision == "f16" and self.f16_layout == "plain" and self.gemm_mode == "split" and D % 64 == 0
                and self.fused_unembed and D <= 8192 and ops.unembed_side_bytes(Cout, Himg, Wimg, kh, kw, sh, sw) > 0)
        h = self._ln(t, self.g_s.norm, f"h{D}", plain=p_ue)
        x_hat = torch.empty((Cout, Himg, Wimg), device=self.device, dtype=torch.float32)
        nside = ops.unembed_side_bytes(Cout, Himg, Wimg, kh, kw, sh, sw) if self.gemm_mode == "split" else 0
        if nside and self.fused_unembed and D <= 8192:
            # ONE fused launch pair: the GEMM epilogue scatters into the reconstruction (de-normalised), the overlap
            # rows go through a 2-rows-per-patch-row side buffer (no [tokens][C*110] column matrix, no overlap-add pass)
            side = self._buf("ue_side", (nside // 4,))
            wue = self._wplain("g_s.final", self.g_s.final.weight) if p_ue else self._wsplit("g_s.final", self.g_s.final.weight)
            ops.gemm_unembed(h, wue, Cout, Himg, Wimg, kh, kw, sh, sw, side,
                             mean=mean, std=std, out=x_hat, hi_only=self.precision == "f16")
            return x_hat
        ncol = Cout * kh * kw
        cols = self._buf("ue_cols", (N, ncol))
        self._mm(h, "g_s.final", self.g_s.final.weight, out=cols)
        ops.col2im(cols, Cout, kh, kw, sh, sw, self.Hp, self.Wp, mean=mean, std=std, out=x_hat)
        return x_hat

    # ---- latent side: everything between y and the entropy coder ---------------------------
    def _latent_side_frame(self, y, want_lik=False, z_sym_out=None):
        z = self._h_a_frame(y)
        med, pk = self.entropy_bottleneck.device_params()
        eb = ops.entropy_bottleneck(med, pk, z=z, want=("sym", "z_hat") + (("lik",) if want_lik else ()),
                                    lik_bound=self.entropy_bottleneck.likelihood_bound, sym_out=z_sym_out)
        scales, means = self._h_s_frame(eb["z_hat"])
        st = self.gaussian_conditional.scale_table
        want = ("idx", "sym", "y_hat") + (("lik",) if want_lik else ())
        if st.numel() == 0:
            want = tuple(w for w in want if w != "idx")
        gc = ops.gaussian_conditional(scales.contiguous(), means.contiguous(), st if st.numel() else None,
                                      y=y.contiguous(), want=want,
                                      scale_bound=self._scale_bound(),
                                      lik_bound=self.gaussian_conditional.likelihood_bound)
        return dict(z=z, z_sym=eb["sym"], z_hat=eb["z_hat"], z_lik=eb.get("lik"), scales=scales, means=means,
                    idx=gc.get("idx"), y_sym=gc["sym"], y_hat=gc["y_hat"], y_lik=gc.get("lik"))

    # ---- GPU phases -------------------------------------------------------------------------
    @contextlib.contextmanager
    def _gpu_phase(self, light=False, prio=1):
        """One frame's GPU phase (a run of kernel launches ended by a stream sync, which the
        device->host hand-off to the entropy coder needs anyway).  When several frames are in
        flight (cra5_amd/pipeline.py) phases of different frames take turns on the GPU at this
        granularity while the other frames sit in their host rANS phase: every kernel runs
        alone on the chip (clean per-kernel timing, no L2 thrash between frames)."""
        log = self.phase_log
        t0 = time.perf_counter() if log is not None else 0.0
        if not self.gpu_exclusive:
            sem = None
            if self.gpu_slots > 0 and not light:
                sem = self._gpu_sem
                if sem is None or sem[0] != self.gpu_slots:
                    with self._gpu_lock:
                        sem = self._gpu_sem
                        if sem is None or sem[0] != self.gpu_slots:
                            sem = self._gpu_sem = (self.gpu_slots, _SlotGate(self.gpu_slots))
                sem[1].acquire(prio)
            t1 = time.perf_counter() if log is not None else 0.0
            try:
                if light:
                    # the ~1 ms h_s phase sits between a frame's two host phases: on a high-priority
                    # HIP stream its small kernels are scheduled ahead of the other frames' queued
                    # blocks instead of behind them (the previous phase of this frame ended with a
                    # stream sync, so switching streams needs no event)
                    hp = getattr(self._tls, "hp_stream", None)
                    if hp is None:
                        hp = self._tls.hp_stream = torch.cuda.Stream(device=self.device, priority=-1)
                    with torch.cuda.stream(hp), ops.stream_scope():
                        yield
                        hp.synchronize()
                else:
                    with ops.stream_scope():   # (the launch stream of this thread, looked up once per phase)
                        yield
                    torch.cuda.current_stream().synchronize()
            finally:
                if sem is not None:
                    sem[1].release()
            if log is not None:
                log.append((threading.get_ident(), t0, t1, time.perf_counter()))
            return
        with self._gpu_lock:
            t1 = time.perf_counter() if log is not None else 0.0
            with ops.stream_scope():
                yield
            torch.cuda.current_stream().synchronize()
            if log is not None:
                log.append((threading.get_ident(), t0, t1, time.perf_counter()))

    def _scale_bound(self):
        """GaussianConditional.lower_bound_scale.bound as a host float, cached: reading a device
        buffer with float() is a full device sync in the middle of a GPU phase."""
        b = self.gaussian_conditional.lower_bound_scale.bound
        key = (b.data_ptr(), b._version)
        c = self.__dict__.get("_sb_cache")
        if c is None or c[0] != key:
            c = (key, float(b.detach().cpu()))
            self.__dict__["_sb_cache"] = c
        return c[1]

    def _pinned(self, name, shape, dtype):
        """Per-thread pinned host staging buffer."""
        ws = getattr(self._tls, "pin", None)
        if ws is None:
            ws = self._tls.pin = {}
        key = (tuple(shape), dtype)
        b = ws.get(name)
        if b is None or b[0] != key:
            b = (key, torch.empty(shape, dtype=dtype, pin_memory=True))
            ws[name] = b
        return b[1]

    def _pack(self, name, fields):
        """One per-thread device byte buffer holding several typed records back to back (16-byte aligned): the phase's
        small results travel to the host as ONE copy.  fields: [(key, numel, dtype)] -> (buffer, {key: typed device view},
        {key: (byte offset, numel, dtype)})."""
        off, lay = 0, {}
        for key, n, dt in fields:
            sz = n * torch.empty((), dtype=dt).element_size()
            lay[key] = (off, n, dt)
            off = (off + sz + 15) // 16 * 16
        buf = self._buf(name, (off,), torch.uint8)
        views = {k: buf[o:o + n * torch.empty((), dtype=dt).element_size()].view(dt) for k, (o, n, dt) in lay.items()}
        return buf, views, lay

    @staticmethod
    def _unpack(host_buf, lay):
        return {k: host_buf[o:o + n * torch.empty((), dtype=dt).element_size()].view(dt) for k, (o, n, dt) in lay.items()}

    def _to_host(self, name, t):
        h = self._pinned(name, t.shape, t.dtype)
        h.copy_(t, non_blocking=True)
        return h


    # ---- public surface (names / return shapes of the reference) ---------------------------
    def _encode_y_guarded(self, x, mean=None, std=None):
        """g_a + quant_conv of one frame under the range guard: x [C, H, W] -> y [L, Hp, Wp]."""
        def run():
            with self._gpu_phase():
                y = self._encode_y_frame(x, mean=mean, std=std)
                flag = self._to_host("ok1", self._probe(y))
            return y, self._finite(flag), None
        return self._range_guard(0, run, "encode")

    def _decode_guarded(self, y_hat, mean=None, std=None):
        """g_s of one frame under the range guard: y_hat [L, Hp, Wp] -> x_hat [C, H, W]."""
        D = self.cfg['embed_dim']

        def run():
            with self._gpu_phase():
                x_hat = self._decode_frame(y_hat, mean=mean, std=std)
                # (the residual stream after the last block carries every upstream poison - token-wise, and the global
                # attention spreads it.  The final LayerNorm's split store is O(gamma * sqrt(D)): a checkpoint with
                # |gamma| sqrt(D) >= 65 504, or a non-finite mean / std, poisons x_hat BEHIND that probe - a strided
                # sample of the reconstruction itself (every 2053rd pixel: ~14 samples of each token's 268 x 110-pixel
                # patch, ~500 of every channel) is probed as well: ADVICE r4)
                flag = self._to_host("ok1", self._probe(self._buf(f"t{D}", (self.Hp * self.Wp, D)), (x_hat, 2053)))
            return x_hat, self._finite(flag), None
        return self._range_guard(1, run, "decode")

    def _latent_side_guarded(self, y, want_lik=False):
        """Hyper-prior + entropy parameters of one frame; the hyper-prior engine is pinned: non-finite -> error."""
        with self._gpu_phase():
            s = self._latent_side_frame(y, want_lik=want_lik)
            flag = self._to_host("ok_h", self._probe(*self._hs_parent(s["scales"], s["means"])))
        if not self._finite(flag):
            raise FloatingPointError("the hyper-prior path produced non-finite entropy parameters (pinned engine, no "
                                     "fallback): the latent handed in is non-finite, or the checkpoint is broken")
        return s

    @torch.no_grad()
    def encode_latent(self, x, type='quantized'):
        """vaeformer.py:272-292 -> (y, y_hat, y_likelihoods)."""
        self._require_gpu()
        ys, yh, yl = [], [], []
        for b in range(x.shape[0]):
            y = self._encode_y_guarded(x[b])
            ys.append(y)
            if type == "quantized":
                s = self._latent_side_guarded(y, want_lik=True)
                yh.append(s["y_hat"].reshape(y.shape))
                yl.append(s["y_lik"].reshape(y.shape))
        y = torch.stack(ys)
        if type == "quantized":
            return y, torch.stack(yh), torch.stack(yl)
        return y, None, None

    @torch.no_grad()
    def decode_latent(self, y, type='quantized'):
        """vaeformer.py:294-300."""
        self._require_gpu()
        return torch.stack([self._decode_guarded(y[b]) for b in range(y.shape[0])])

    @torch.no_grad()
    def forward(self, x):
        """vaeformer.py:302-333."""
        self._require_gpu()
        xh, yl, zl, ys = [], [], [], []
        for b in range(x.shape[0]):
            y = self._encode_y_guarded(x[b])
            s = self._latent_side_guarded(y, want_lik=True)
            xh.append(self._decode_guarded(s["y_hat"].reshape(y.shape)))
            yl.append(s["y_lik"].reshape(y.shape))
            zl.append(s["z_lik"].reshape(-1, self.Hz, self.Wz))
            ys.append(y)
        return {"x_hat": torch.stack(xh), "likelihoods": {"y": torch.stack(yl), "z": torch.stack(zl)},
                "posterior": _Posterior(torch.stack(ys))}

    def _z_pool(self):
        p = self.__dict__.get("_zpool")
        if p is None:
            with self._derive_lock:
                p = self.__dict__.get("_zpool")
                if p is None:
                    from concurrent.futures import ThreadPoolExecutor
                    p = self.__dict__["_zpool"] = ThreadPoolExecutor(max_workers=4, thread_name_prefix="cra5-z")
        return p

    def _encode_z(self, z_np, size):
        eb = self.entropy_bottleneck
        return eb.encode_symbols(z_np, eb._build_indexes(size))

    def _compress_frame(self, x=None, y=None, mean=None, std=None):
        """GPU phase (g_a + latent side, symbols staged to pinned host memory) followed by the
        host phase (two rANS streams).  Either x [C,H,W] or y [L,Hp,Wp]."""
        self.entropy_bottleneck._check()
        self.gaussian_conditional._check()
        gc = self.gaussian_conditional

        P = ops.PROBE_PARTIALS

        def gpu_side():
            with self._gpu_phase(prio=0):
                yy = y if y is not None else self._encode_y_frame(x, mean=mean, std=std)
                if self.resolve_on_gpu and self.compact_records:
                    # Everything the host phase needs leaves the device as ONE copy (round 5; rounds 1-4: five): the z
                    # symbols, the y records - symbol -> (start | range, 16-bit escape record) resolved against the CDF
                    # tables on the device (the host coder is a pure state-update loop, 6 instead of 9 bytes
                    # per latent cross PCIe) - their overflow word and the finiteness probes of y and of mu / sigma are
                    # written by their kernels into slices of one per-thread record buffer.
                    nz, nl = self.entropy_bottleneck.channels * self.Hz * self.Wz, yy.numel()
                    buf, v, lay = self._pack("enc_pack", [("z_sym", nz, torch.int32), ("sr", nl, torch.int32),
                                                           ("rec", nl, torch.int16), ("ovf", 1, torch.int32),
                                                           ("probe", 3 * P, torch.float32)])
                    s = self._latent_side_frame(yy.contiguous(), z_sym_out=v["z_sym"])
                    par = self._hs_parent(s["scales"], s["means"])
                    self._probe(yy, *par, out=v["probe"][:(1 + len(par)) * P])
                    ops.rans_resolve_symbols_compact(s["y_sym"].reshape(-1), s["idx"].reshape(-1), gc._quantized_cdf,
                                                     gc._cdf_length, gc._offset, out=(v["sr"], v["rec"], v["ovf"]))
                    keep["sym"], keep["idx"] = s["y_sym"], s["idx"]   # for the (rare) 32-bit re-resolve
                    h = self._unpack(self._to_host("enc_pack", buf), lay)
                    fl = h["probe"][:(1 + len(par)) * P]
                    z_sym, host = h["z_sym"].view(tuple(s["z_sym"].shape)), ("compact", h["sr"], h["rec"], h["ovf"])
                else:
                    s = self._latent_side_frame(yy.contiguous())
                    par = self._hs_parent(s["scales"], s["means"])
                    pr = self._probe(yy, *par)          # [y | mu, sigma]: PROBE_PARTIALS values each
                    z_sym = self._to_host("z_sym", s["z_sym"])
                    if self.resolve_on_gpu:
                        sr, raw, esc = ops.rans_resolve_symbols(s["y_sym"].reshape(-1), s["idx"].reshape(-1),
                                                                gc._quantized_cdf, gc._cdf_length, gc._offset)
                        host = ("resolved", self._to_host("y_sr", sr), self._to_host("y_raw", raw), self._to_host("y_esc", esc))
                    else:
                        host = ("plain", self._to_host("y_sym", s["y_sym"]), self._to_host("idx", s["idx"]))
                    fl = self._to_host("enc_flags", pr)
            return (z_sym, host), self._finite(fl[:P]), self._finite(fl[P:])   # (the phase ended with a stream sync)
        keep = {}
        z_sym, host = self._range_guard(0, gpu_side, "compress")
        t_host = time.perf_counter()
        # the z stream is coded beside the y stream (round 6): two independent coders, the native calls release the GIL -
        # 2 ms off the serial host phase of every frame (16.5 ms with the default synthetic weights, 8.5 entropy-matched)
        z_job = self._z_pool().submit(self._encode_z, z_sym.numpy().reshape(-1), (1, z_sym.shape[0], z_sym.shape[1]))   # (joined below)
        try:
            if host[0] == "compact" and int(host[3][0]) != 0:
                # an escape payload beyond 12 bits (|symbol| thousands beyond its table row): this frame takes the 32-bit records
                with self._gpu_phase(light=True):
                    sr, raw, esc = ops.rans_resolve_symbols(keep["sym"].reshape(-1), keep["idx"].reshape(-1),
                                                            gc._quantized_cdf, gc._cdf_length, gc._offset)
                    host = ("resolved", self._to_host("y_sr", sr), self._to_host("y_raw", raw), self._to_host("y_esc", esc))
            keep.clear()
            if host[0] == "compact":
                rec_np = host[2].numpy()
                y_str = ops.rans_encode_resolved_compact(host[1].numpy(), rec_np)
                n_esc = int(np.count_nonzero(rec_np))
            elif host[0] == "resolved":
                esc_np = host[3].numpy()
                y_str = ops.rans_encode_resolved(host[1].numpy(), host[2].numpy(), esc_np)
                n_esc = int(np.count_nonzero(esc_np))
            else:
                sym_np, idx_np = host[1].numpy().reshape(-1), host[2].numpy().reshape(-1)
                y_str = gc.encode_symbols(sym_np, idx_np)
                _, ln, off = gc.host_tables()
                v = sym_np - off[idx_np]
                n_esc = int(np.count_nonzero((v < 0) | (v >= ln[idx_np] - 2)))
        except BaseException:
            z_job.cancel()
            try:
                z_job.result()          # (the z coder reads this thread's pinned record buffer: let it finish before we leave)
            except BaseException:  # noqa: BLE001
                pass
            raise
        z_str = z_job.result()
        # symbols of the y stream coded through the escape path (rans_interface.cpp:120-160): the SURVEY 8(e) stats field
        self._tls.last_n_escape = n_esc
        if self.host_log is not None:
            self.host_log.append(("enc", time.perf_counter() - t_host))
        return y_str, z_str

    @torch.no_grad()
    def compress_from_latent(self, y):
        """vaeformer.py:334-348."""
        self._require_gpu()
        ystr, zstr, nesc = [], [], []
        for b in range(y.shape[0]):
            a, c = self._compress_frame(y=y[b])
            ystr.append(a)
            zstr.append(c)
            nesc.append(self._tls.last_n_escape)
        self._tls.n_escape = nesc     # (not a dict key: the returned dict has exactly the reference's two)
        return {"strings": [ystr, zstr], "z_shape": torch.Size([self.Hz, self.Wz])}

    @torch.no_grad()
    def compress(self, x):
        """vaeformer.py:350-376."""
        self._require_gpu()
        ystr, zstr, nesc = [], [], []
        for b in range(x.shape[0]):
            a, c = self._compress_frame(x=x[b])
            ystr.append(a)
            zstr.append(c)
            nesc.append(self._tls.last_n_escape)
        self._tls.n_escape = nesc     # (not a dict key: the returned dict has exactly the reference's two)
        return {"strings": [ystr, zstr], "z_shape": torch.Size([self.Hz, self.Wz])}

    def last_n_escape(self):
        """Escape-coded symbols of each y stream the calling thread's last compress() / compress_from_latent() wrote
        (rans_interface.cpp:120-160): the `n_escape` field of the per-frame stats row."""
        return list(getattr(self._tls, "n_escape", []))

    @staticmethod
    def _y_desync(e):
        """The y stream decoded to its end with the wrong final coder state.  The decoder re-derives every CDF index
        from ITS OWN h_s(z_hat) (vaeformer.py:378-400 in the reference): one index that differs from the encoder's -
        h_s evaluated by another platform's float arithmetic, DESIGN.md section 2 - desynchronises everything after
        it.  The reference decodes such a stream into a plausible-looking wrong frame; here it is an error."""
        return StreamDesyncError(
            "decompress: the y stream does not end in the coder's initial state.  The z stream decoded cleanly, so "
            "the most likely cause is a hyper-prior CDF-index mismatch with the ENCODER's platform: this build's "
            "h_s(z_hat) rounds at least one scale into a different table row than the build that wrote the stream "
            "(a .bin written by the PyTorch reference or by another engine; DESIGN.md section 2).  A damaged y "
            "stream or a different checkpoint looks the same.  Nothing was reconstructed", e.status)

    def _decompress_frame(self, y_string, z_string, shape, reconstruct, mean=None, std=None):
        """host: decode z | GPU: h_s, indexes | host: decode y | GPU: de-quantise (+ g_s)."""
        eb, gc = self.entropy_bottleneck, self.gaussian_conditional
        Cz = eb.channels
        zh, zw = int(shape[0]), int(shape[1])
        z_idx = eb._build_indexes((1, Cz, zh, zw))
        z_host = self._pinned("z_in", (Cz, zh * zw), torch.int32)
        t_host = time.perf_counter()
        try:
            eb.decode_symbols(z_string, z_idx, out=z_host.numpy().reshape(-1))   # straight into pinned memory
        except StreamDesyncError as e:
            raise StreamDesyncError(
                "decompress: the z stream does not end in the coder's initial state - it was not written with this "
                "checkpoint's EntropyBottleneck tables (update() not run on the same weights?) or it is damaged; "
                "nothing was reconstructed", e.status) from e
        if self.host_log is not None:
            self.host_log.append(("dec_z", time.perf_counter() - t_host))
        with self._gpu_phase(light=True):
            z_sym = z_host.to(self.device, non_blocking=True)
            med, _ = eb.device_params()
            z_hat = ops.entropy_bottleneck(med, None, sym_in=z_sym, want=("z_hat",))["z_hat"]
            scales, means = self._h_s_frame(z_hat)
            scales, means = scales.contiguous(), means.contiguous()
            compact = self.compact_records and gc.scale_table.numel() <= 256
            par = self._hs_parent(scales, means)
            if compact:
                # compact records (round 4): uint8 CDF indexes to the host, int16 symbols back - 8 instead of 21 MB of
                # PCIe traffic on the decode side's latency path; the indexes and the finiteness probes of mu / sigma
                # leave as ONE copy (round 5)
                P = ops.PROBE_PARTIALS
                buf, v, lay = self._pack("dec_pack", [("idx8", means.numel(), torch.uint8), ("probe", 2 * P, torch.float32)])
                ops.gaussian_conditional_compact(scales, means, gc.scale_table, want_idx8=True,
                                                 scale_bound=self._scale_bound(), idx8_out=v["idx8"])
                self._probe(*par, out=v["probe"][:len(par) * P])
                h = self._unpack(self._to_host("dec_pack", buf), lay)
                idx_h, ok_h = h["idx8"], h["probe"][:len(par) * P]
            else:
                idx = ops.gaussian_conditional(scales, means, gc.scale_table,
                                               sym_in=torch.zeros_like(means, dtype=torch.int32), want=("idx",),
                                               scale_bound=self._scale_bound())["idx"]
                idx_h = self._to_host("idx", idx)
                ok_h = self._to_host("ok_h", self._probe(*par))
        if not self._finite(ok_h):
            raise FloatingPointError("decompress: the hyper-prior path produced non-finite entropy parameters (pinned "
                                     "engine, no fallback): the stream does not belong to this checkpoint, or the "
                                     "checkpoint is broken")
        y_host = None
        t_host = time.perf_counter()
        if compact:
            y_host = self._pinned("y_in16", tuple(means.shape), torch.int16)
            try:
                gc.decode_symbols_compact(y_string, idx_h.numpy().reshape(-1), y_host.numpy().reshape(-1))
            except StreamDesyncError as e:
                raise self._y_desync(e) from e
            except Cra5Error as e:
                if e.status != ERR_RANGE:
                    raise
                compact = False                      # a symbol beyond int16: the 32-bit records decode the same stream
                idx_h = idx_h.to(torch.int32)
        if not compact:
            y_host = self._pinned("y_in", tuple(means.shape), torch.int32)
            try:
                gc.decode_symbols(y_string, idx_h.numpy().reshape(-1), out=y_host.numpy().reshape(-1))
            except StreamDesyncError as e:
                raise self._y_desync(e) from e
        if self.host_log is not None:
            self.host_log.append(("dec_y", time.perf_counter() - t_host))

        def gpu_side():
            with self._gpu_phase(prio=2):
                y_sym = y_host.to(self.device, non_blocking=True)
                if compact:
                    y_hat = ops.gaussian_conditional_compact(None, means, sym16_in=y_sym)["y_hat"]
                else:
                    y_hat = ops.gaussian_conditional(scales, means, gc.scale_table, sym_in=y_sym, want=("y_hat",))["y_hat"]
                if not reconstruct:
                    return y_hat, True, None
                x_hat = self._decode_frame(y_hat, mean=mean, std=std)
                flag = self._to_host("ok1", self._probe(self._buf(f"t{self.cfg['embed_dim']}",
                                                                  (self.Hp * self.Wp, self.cfg['embed_dim'])),
                                                        (x_hat, 2053)))
            return x_hat, self._finite(flag), None
        return self._range_guard(1, gpu_side, "decompress")

    @torch.no_grad()
    def decompress(self, strings, shape, return_format='reconstructed'):
        """vaeformer.py:378-400."""
        self._require_gpu()
        assert isinstance(strings, list) and len(strings) == 2
        self.entropy_bottleneck._check()
        self.gaussian_conditional._check()
        rec = return_format != 'latent'
        out = torch.stack([self._decompress_frame(strings[0][b], strings[1][b], shape, rec)
                           for b in range(len(strings[0]))])
        if not rec:
            return out
        return {"x_hat": out}

    @torch.no_grad()
    def prediction(self, inputs):
        """vaeformer.py:254-269 (the reference reads out['shape'], a key compress() never
        sets; we use 'z_shape')."""
        import time
        t1 = time.time()
        out = self.compress(inputs)
        torch.cuda.synchronize()
        t2 = time.time()
        x_hat = self.decompress(out['strings'], out['z_shape'])
        torch.cuda.synchronize()
        t3 = time.time()
        return {**x_hat, "strings": out['strings'], "z_shape": out['z_shape'], 'x_shape': inputs.shape,
                'encoding_time': (t2 - t1) / inputs.size(0), 'decoding_time': (t3 - t2) / inputs.size(0)}

    def get_last_layer(self):
        return self.g_s.final.weight
