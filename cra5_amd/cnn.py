"""CNN zoo codecs on the MI355X kernels: `bmshj2018-factorized`, `bmshj2018-factorized-relu`, `bmshj2018-hyperprior`,
`mbt2018-mean` (FactorizedPrior / FactorizedPriorReLU / ScaleHyperprior / MeanScaleHyperprior of the reference,
cra5/models/compressai/models/google.py:64-508) - SURVEY.md section 8(f)-4.

Same module tree (state-dict keys `g_a.0.weight`, `g_a.1.beta`, ..., `h_s.4.bias`, entropy-model
buffers), same forward / compress / decompress results as the reference classes; every tensor op
is a HIP kernel behind the C ABI: conv5x5-s2 / conv3x3 = padded patch gather + split-f16 GEMM,
deconv5x5-s2 = GEMM + deterministic gather overlap-add, GDN / IGDN (`cra5_gdn_f32`), ReLU /
LeakyReLU / abs, the EntropyBottleneck / GaussianConditional kernels and the rANS coder of the
VAEformer path.  These are the reference's comparison baselines, not the ERA5 hot path: the kernels
are the simple one-thread-per-element forms, images are processed one at a time.  No CPU fallback.
"""
import torch
import torch.nn as nn

from . import ops
from .entropy import EntropyBottleneck, GaussianConditional, get_scale_table
from .layers import GDN

__all__ = ["FactorizedPrior", "FactorizedPriorReLU", "ScaleHyperprior", "MeanScaleHyperprior", "cnn_model", "CNN_CFGS"]


class _Conv(nn.Module):
    """Parameters of models/utils.py:128-135 `conv` (nn.Conv2d(k, stride, padding=k//2))."""

    def __init__(self, cin, cout, kernel_size=5, stride=2):
        super().__init__()
        self.k, self.stride = kernel_size, stride
        m = nn.Conv2d(cin, cout, kernel_size, stride, kernel_size // 2)     # parameter shapes / default init
        self.weight, self.bias = m.weight, m.bias
        self._ws = None

    def forward(self, x):
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device))
        if self._ws is None or self._ws[0] != key:
            self._ws = (key, ops.split_f16(self.weight.detach().reshape(self.weight.shape[0], -1).contiguous(), "auto"))
        return ops.conv2d(x, self._ws[1], self.bias, self.k, self.stride)


class _Deconv(nn.Module):
    """models/utils.py:138-146 `deconv` (nn.ConvTranspose2d(k, stride, output_padding=stride-1, padding=k//2))."""

    def __init__(self, cin, cout, kernel_size=5, stride=2):
        super().__init__()
        self.k, self.stride, self.cout = kernel_size, stride, cout
        m = nn.ConvTranspose2d(cin, cout, kernel_size, stride, kernel_size // 2, stride - 1)
        self.weight, self.bias = m.weight, m.bias
        self._ws = None

    def forward(self, x):
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device))
        if self._ws is None or self._ws[0] != key:
            w2 = self.weight.detach().reshape(self.weight.shape[0], -1).t().contiguous()   # [(co,i,j)][ci]
            self._ws = (key, ops.split_f16(w2, "auto"))
        return ops.conv_transpose2d(x, self._ws[1], self.bias, self.cout, self.k, self.stride)


class _Act(nn.Module):
    def __init__(self, op):
        super().__init__()
        self.op = op

    def forward(self, x):
        return ops.unary(x, self.op)


class _GDN3(GDN):
    """GDN on one image [C, H, W]."""

    def forward(self, x):
        return super().forward(x.unsqueeze(0))[0]


def _seq(*mods):
    return nn.Sequential(*mods)


class _CnnBase(nn.Module):
    def _require_gpu(self, x):
        if not x.is_cuda:
            raise RuntimeError("cra5_amd CNN codecs compute only on an MI355X (HIP kernels, no CPU fallback)")

    def update(self, scale_table=None, force=False):
        """CompressionModel.update (models/base.py:91-115)."""
        updated = self.entropy_bottleneck.update(force=force)
        gc = getattr(self, "gaussian_conditional", None)
        if gc is not None:
            updated |= gc.update_scale_table(get_scale_table() if scale_table is None else scale_table, force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        for name in ("entropy_bottleneck", "gaussian_conditional"):
            mod = getattr(self, name, None)
            for b in ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"):
                key = f"{name}.{b}"
                if mod is not None and key in state_dict and getattr(mod, b).numel() == 0:
                    getattr(mod, b).resize_(state_dict[key].size())
        return nn.Module.load_state_dict(self, state_dict, strict=strict)

    @classmethod
    def from_state_dict(cls, state_dict):
        net = cls(state_dict["g_a.0.weight"].size(0), state_dict["g_a.6.weight"].size(0))
        net.load_state_dict(state_dict)
        return net

    # EntropyBottleneck on one [C, H, W] tensor
    def _eb(self, z, want):
        med, pk = self.entropy_bottleneck.device_params()
        C = z.shape[0]
        return ops.entropy_bottleneck(med, pk, z=z.reshape(C, -1).contiguous(), want=want,
                                      lik_bound=self.entropy_bottleneck.likelihood_bound)

    def _eb_dequant(self, sym, C):
        med, _ = self.entropy_bottleneck.device_params()
        return ops.entropy_bottleneck(med, None, sym_in=sym.reshape(C, -1).contiguous(), want=("z_hat",))["z_hat"]


class FactorizedPrior(_CnnBase):
    """google.py:64-164 (`bmshj2018-factorized`)."""

    def __init__(self, N, M, in_channel=3, **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(M)
        self.g_a = _seq(_Conv(in_channel, N), _GDN3(N), _Conv(N, N), _GDN3(N), _Conv(N, N), _GDN3(N), _Conv(N, M))
        self.g_s = _seq(_Deconv(M, N), _GDN3(N, inverse=True), _Deconv(N, N), _GDN3(N, inverse=True), _Deconv(N, N),
                        _GDN3(N, inverse=True), _Deconv(N, in_channel))
        self.N, self.M = int(N), int(M)
        self.eval()

    @property
    def downsampling_factor(self):
        return 2 ** 4

    @torch.no_grad()
    def forward(self, x):
        self._require_gpu(x)
        xs, ls = [], []
        for b in range(x.shape[0]):
            y = self.g_a(x[b])
            e = self._eb(y, ("z_hat", "lik"))
            xs.append(self.g_s(e["z_hat"].reshape(y.shape)))
            ls.append(e["lik"].reshape(y.shape))
        return {"x_hat": torch.stack(xs), "likelihoods": {"y": torch.stack(ls)}}

    @torch.no_grad()
    def compress(self, x):
        self._require_gpu(x)
        self.entropy_bottleneck._check()
        strings, shape = [], None
        for b in range(x.shape[0]):
            y = self.g_a(x[b])
            shape = y.shape[-2:]
            sym = self._eb(y, ("sym",))["sym"].cpu().numpy().reshape(-1)
            idx = self.entropy_bottleneck._build_indexes((1,) + tuple(y.shape))
            strings.append(self.entropy_bottleneck.encode_symbols(sym, idx))
        return {"strings": [strings], "shape": torch.Size(shape)}

    @torch.no_grad()
    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 1
        dev = self.g_a[0].weight.device
        out = []
        for s in strings[0]:
            idx = self.entropy_bottleneck._build_indexes((1, self.M, int(shape[0]), int(shape[1])))
            sym = torch.from_numpy(self.entropy_bottleneck.decode_symbols(s, idx)).to(dev)
            y_hat = self._eb_dequant(sym, self.M).reshape(self.M, int(shape[0]), int(shape[1]))
            out.append(self.g_s(y_hat))
        return {"x_hat": torch.stack(out)}


class FactorizedPriorReLU(FactorizedPrior):
    """google.py:166-199 (`bmshj2018-factorized-relu`): the GDN / IGDN stages replaced by ReLU."""

    def __init__(self, N, M, rate_distortion_loss=None, in_channel=3, **kwargs):
        super().__init__(N, M, in_channel=in_channel)
        self.g_a = _seq(_Conv(in_channel, N), _Act("relu"), _Conv(N, N), _Act("relu"), _Conv(N, N), _Act("relu"), _Conv(N, M))
        self.g_s = _seq(_Deconv(M, N), _Act("relu"), _Deconv(N, N), _Act("relu"), _Deconv(N, N), _Act("relu"),
                        _Deconv(N, in_channel))
        self.eval()


class ScaleHyperprior(_CnnBase):
    """google.py:227-383 (`bmshj2018-hyperprior`): zero-mean Gaussian conditional, scales from h_s(z_hat)."""

    _act = "relu"
    _mean = False

    def __init__(self, N, M, rate_distortion_loss=None, in_channel=3, **kwargs):
        super().__init__()
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.g_a = _seq(_Conv(in_channel, N), _GDN3(N), _Conv(N, N), _GDN3(N), _Conv(N, N), _GDN3(N), _Conv(N, M))
        self.g_s = _seq(_Deconv(M, N), _GDN3(N, inverse=True), _Deconv(N, N), _GDN3(N, inverse=True), _Deconv(N, N),
                        _GDN3(N, inverse=True), _Deconv(N, in_channel))
        self._build_hyper(N, M)
        self.gaussian_conditional = GaussianConditional(None)
        self.N, self.M = int(N), int(M)
        self.eval()

    def _build_hyper(self, N, M):
        self.h_a = _seq(_Conv(M, N, 3, 1), _Act("relu"), _Conv(N, N), _Act("relu"), _Conv(N, N))
        self.h_s = _seq(_Deconv(N, N), _Act("relu"), _Deconv(N, N), _Act("relu"), _Conv(N, M, 3, 1), _Act("relu"))

    @property
    def downsampling_factor(self):
        return 2 ** (4 + 2)

    def _params(self, z_hat):
        """h_s(z_hat) -> (scales, means or zeros), each [M, H, W] (google.py:349-350 / 472-474)."""
        p = self.h_s(z_hat)
        if self._mean:
            return p[: self.M].contiguous(), p[self.M:].contiguous()
        return p.contiguous(), torch.zeros_like(p)

    def _h_a_in(self, y):
        return y if self._mean else ops.unary(y, "abs")          # google.py:347 torch.abs(y) | :469 y

    def _side(self, y, want):
        z = self.h_a(self._h_a_in(y))
        e = self._eb(z, ("sym", "z_hat", "lik"))
        scales, means = self._params(e["z_hat"].reshape(z.shape))
        gc = self.gaussian_conditional
        st = gc.scale_table if gc.scale_table.numel() else None
        if st is None:
            want = tuple(w for w in want if w != "idx")
        g = ops.gaussian_conditional(scales, means, st, y=y.contiguous(), want=want,
                                     scale_bound=float(gc.lower_bound_scale.bound), lik_bound=gc.likelihood_bound)
        return z, e, g

    @torch.no_grad()
    def forward(self, x):
        self._require_gpu(x)
        xs, ly, lz = [], [], []
        for b in range(x.shape[0]):
            y = self.g_a(x[b])
            z, e, g = self._side(y, ("y_hat", "lik"))
            xs.append(self.g_s(g["y_hat"].reshape(y.shape)))
            ly.append(g["lik"].reshape(y.shape))
            lz.append(e["lik"].reshape(z.shape))
        return {"x_hat": torch.stack(xs), "likelihoods": {"y": torch.stack(ly), "z": torch.stack(lz)}}

    @torch.no_grad()
    def compress(self, x):
        self._require_gpu(x)
        self.entropy_bottleneck._check()
        self.gaussian_conditional._check()
        ys, zs, shape = [], [], None
        for b in range(x.shape[0]):
            y = self.g_a(x[b])
            z, e, g = self._side(y, ("idx", "sym"))
            shape = z.shape[-2:]
            z_idx = self.entropy_bottleneck._build_indexes((1,) + tuple(z.shape))
            zs.append(self.entropy_bottleneck.encode_symbols(e["sym"].cpu().numpy().reshape(-1), z_idx))
            ys.append(self.gaussian_conditional.encode_symbols(g["sym"].cpu().numpy().reshape(-1),
                                                               g["idx"].cpu().numpy().reshape(-1)))
        return {"strings": [ys, zs], "shape": torch.Size(shape)}

    @torch.no_grad()
    def decompress(self, strings, shape):
        assert isinstance(strings, list) and len(strings) == 2
        dev = self.g_a[0].weight.device
        gc = self.gaussian_conditional
        zh, zw = int(shape[0]), int(shape[1])
        out = []
        for y_s, z_s in zip(strings[0], strings[1]):
            z_idx = self.entropy_bottleneck._build_indexes((1, self.N, zh, zw))
            z_sym = torch.from_numpy(self.entropy_bottleneck.decode_symbols(z_s, z_idx)).to(dev)
            z_hat = self._eb_dequant(z_sym, self.N).reshape(self.N, zh, zw)
            scales, means = self._params(z_hat)
            idx = ops.gaussian_conditional(scales, means, gc.scale_table, sym_in=torch.zeros_like(means, dtype=torch.int32),
                                           want=("idx",), scale_bound=float(gc.lower_bound_scale.bound))["idx"]
            y_sym = torch.from_numpy(gc.decode_symbols(y_s, idx.cpu().numpy().reshape(-1))).to(dev)
            y_hat = ops.gaussian_conditional(scales, means, gc.scale_table, sym_in=y_sym.reshape(means.shape).contiguous(),
                                             want=("y_hat",))["y_hat"]
            out.append(self.g_s(y_hat.reshape(means.shape)))
        return {"x_hat": torch.stack(out)}


class MeanScaleHyperprior(ScaleHyperprior):
    """google.py:386-506 (`mbt2018-mean`): h_s emits (scales, means) = chunk(2, 1), LeakyReLU hyper nets."""

    _mean = True

    def _build_hyper(self, N, M):
        self.h_a = _seq(_Conv(M, N, 3, 1), _Act("leaky_relu"), _Conv(N, N), _Act("leaky_relu"), _Conv(N, N))
        self.h_s = _seq(_Deconv(N, M), _Act("leaky_relu"), _Deconv(M, M * 3 // 2), _Act("leaky_relu"),
                        _Conv(M * 3 // 2, M * 2, 3, 1))


# (N, M) per quality, zoo/image.py:202-245
CNN_CFGS = {
    "bmshj2018-factorized": {q: (128, 192) if q <= 5 else (192, 320) for q in range(1, 9)},
    "bmshj2018-factorized-relu": {q: (128, 192) if q <= 5 else (192, 320) for q in range(1, 9)},
    "bmshj2018-hyperprior": {q: (128, 192) if q <= 5 else (192, 320) for q in range(1, 9)},
    "mbt2018-mean": {q: (128, 192) if q <= 4 else (192, 320) for q in range(1, 9)},
}
_ARCH = {"bmshj2018-factorized": FactorizedPrior, "bmshj2018-factorized-relu": FactorizedPriorReLU,
         "bmshj2018-hyperprior": ScaleHyperprior,
         "mbt2018-mean": MeanScaleHyperprior}


def cnn_model(architecture, quality, metric="mse", pretrained=False, **kwargs):
    """zoo/image.py:248-299 `_load_model`: same errors; weights are never downloaded (no network)."""
    if architecture not in _ARCH:
        raise ValueError(f'Invalid architecture name "{architecture}"')
    if metric not in ("mse", "ms-ssim"):
        raise ValueError(f'Invalid metric "{metric}"')
    if quality not in CNN_CFGS[architecture]:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
    if pretrained:
        raise RuntimeError("Pre-trained model not yet available (no network access)")
    N, M = CNN_CFGS[architecture][quality]
    return _ARCH[architecture](N, M, **kwargs)
