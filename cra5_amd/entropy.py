"""Host side of the two entropy models: CDF-table construction (`update()`), parameter
packing for the device kernels, and the compress / decompress plumbing around the
native rANS coder.

Mirrors the reference's classes by name and behaviour
(cra5/models/compressai/entropy_models/entropy_models.py in taohan10200/CRA5):
EntropyBottleneck (:333-542) and GaussianConditional (:545-685), including their
registered buffers (`_offset`, `_quantized_cdf`, `_cdf_length`, `scale_table`, ...), so a
reference checkpoint loads unchanged.  Table construction runs on the host with
torch-CPU float32 ops + scipy `norm.ppf`, exactly like the reference, never in a GPU
kernel - tables must be bit-identical on the encode and the decode side.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64  # models/base.py:54-56


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    """models/base.py:59-61."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def pmf_to_quantized_cdf(pmf, precision=16):
    """entropy_models.py:89-92, through the C ABI instead of compressai._CXX."""
    return torch.from_numpy(ops.pmf_to_quantized_cdf(pmf.detach().cpu().float().numpy(), precision)
                            .astype(np.int64)).int()


class _LowerBound(nn.Module):
    """Forward of compressai.ops.LowerBound (ops/bound_ops.py:36-80): max(x, bound).
    Kept as a module only for its `bound` buffer (state-dict compatibility)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return torch.max(x, self.bound)


class EntropyModel(nn.Module):
    """entropy_models.py:99-330 (inference subset)."""

    def __init__(self, likelihood_bound=1e-9, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.likelihood_bound = float(likelihood_bound)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = _LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._host_tables = None

    # -- table access ----------------------------------------------------------------
    # (buffer, name in the "Uninitialized ..." message, name in the "Invalid ... size" message, rank);
    # the messages are the reference's (entropy_models.py:218-237), incl. its "offsets" for the lengths
    _TABLES = (("_quantized_cdf", "CDFs", "CDF", 2), ("_offset", "offsets", "offsets", 1),
               ("_cdf_length", "CDF lengths", "offsets", 1))

    def _check(self):
        """Raise the reference's ValueErrors when update() has not run / a table is mis-shaped."""
        for attr, what, what_size, rank in self._TABLES:
            t = getattr(self, attr)
            if t.numel() == 0:
                raise ValueError(f"Uninitialized {what}. Run update() first")
            if t.dim() != rank:
                raise ValueError(f"Invalid {what_size} size {t.size()}")

    def host_tables(self):
        """(cdf, cdf_length, offset) as contiguous int32 numpy arrays, cached: the
        reference re-marshals the 64x3133 table into Python lists on every call
        (entropy_models.py:267-269)."""
        self._check()
        key = (self._quantized_cdf.data_ptr(), self._quantized_cdf._version, self._quantized_cdf.shape)
        if self._host_tables is None or self._host_tables[0] != key:
            self._host_tables = (key,
                                 np.ascontiguousarray(self._quantized_cdf.detach().cpu().numpy(), dtype=np.int32),
                                 np.ascontiguousarray(self._cdf_length.detach().cpu().numpy(), dtype=np.int32),
                                 np.ascontiguousarray(self._offset.detach().cpu().numpy(), dtype=np.int32))
        return self._host_tables[1:]

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """entropy_models.py:208-216."""
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    # -- coding ----------------------------------------------------------------------
    def encode_symbols(self, symbols, indexes):
        """symbols / indexes: int32 host arrays of ONE batch item, (C,H,W) row-major
        (entropy_models.py:263-271) -> bytes."""
        cdf, length, offset = self.host_tables()
        return ops.rans_encode(symbols, indexes, cdf, length, offset)

    def decode_symbols(self, string, indexes, out=None):
        cdf, length, offset = self.host_tables()
        return ops.rans_decode(string, indexes, cdf, length, offset, out=out)

    def decode_symbols_compact(self, string, indexes_u8, out_i16):
        """The same decoder on the frame path's compact records (uint8 indexes in, int16 symbols out)."""
        cdf, length, offset = self.host_tables()
        return ops.rans_decode_compact(string, indexes_u8, cdf, length, offset, out_i16)


def eb_pack_params(sd, prefix="entropy_bottleneck"):
    """Per-channel parameter block for the EB likelihood kernel (58 floats/channel, see
    csrc/elementwise.hip): softplus(matrix) and tanh(factor) are evaluated ONCE here with
    torch-CPU float32, as the reference does on every call (entropy_models.py:434-453)."""
    g = lambda n: sd[f"{prefix}.{n}"].detach().cpu().float()  # noqa: E731
    C = g("_matrix0").shape[0]
    parts = []
    for i in range(5):
        parts.append(F.softplus(g(f"_matrix{i}")).reshape(C, -1))
        parts.append(g(f"_bias{i}").reshape(C, -1))
        if i < 4:
            parts.append(torch.tanh(g(f"_factor{i}")).reshape(C, -1))
    out = torch.cat(parts, 1).contiguous()
    assert out.shape[1] == 58
    return out


class EntropyBottleneck(EntropyModel):
    """entropy_models.py:333-542."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        """Parameters of the factorised density (names / shapes / initial values / registration order
        of entropy_models.py:346-388, so a reference checkpoint loads unchanged): a chain of
        len(filters)+1 per-channel affine maps 1 -> f0 -> ... -> 1 with gated tanh between them."""
        super().__init__(*args, **kwargs)
        C = self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale, self.tail_mass = float(init_scale), float(tail_mass)
        widths = (1,) + self.filters + (1,)
        n_maps = len(widths) - 1
        per_map_scale = self.init_scale ** (1.0 / n_maps)
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            # softplus^-1 of the per-map gain, so that the composed maps start at `init_scale`
            m0 = float(np.log(np.expm1(1.0 / per_map_scale / fan_out)))
            self.register_parameter(f"_matrix{i}", nn.Parameter(torch.full((C, fan_out, fan_in), m0)))
            self.register_parameter(f"_bias{i}", nn.Parameter(torch.empty(C, fan_out, 1).uniform_(-0.5, 0.5)))
            if i + 1 < n_maps:
                self.register_parameter(f"_factor{i}", nn.Parameter(torch.zeros(C, fan_out, 1)))
        q0 = torch.tensor([-self.init_scale, 0.0, self.init_scale])
        self.quantiles = nn.Parameter(q0.repeat(C, 1, 1))
        logit_tail = float(np.log(2.0 / self.tail_mass - 1.0))
        self.register_buffer("target", torch.tensor([-logit_tail, 0.0, logit_tail]))
        self._packed = None

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def _cpu_params(self):
        return {k: v.detach().cpu().float() for k, v in self.named_parameters()}

    def _logits_cumulative_cpu(self, inputs, p):
        """entropy_models.py:434-453 on the host (table construction only)."""
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(p[f"_matrix{i:d}"]), logits)
            logits = logits + p[f"_bias{i:d}"]
            if i < len(self.filters):
                logits = logits + torch.tanh(p[f"_factor{i:d}"]) * torch.tanh(logits)
        return logits

    @torch.no_grad()
    def update(self, force=False):
        """entropy_models.py:394-427."""
        if self._offset.numel() > 0 and not force:
            return False
        p = self._cpu_params()
        q = p["quantiles"]
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max().item()
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative_cpu(samples - 0.5, p)
        upper = self._logits_cumulative_cpu(samples + 0.5, p)
        pmf = (torch.sigmoid(upper) - torch.sigmoid(lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        dev = self._offset.device
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-minima).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._host_tables = None
        return True

    def device_params(self):
        """(medians [C], packed likelihood params [C,58]) on the module's device, cached."""
        ver = tuple(p._version for p in self.parameters()) + (str(self.quantiles.device),)
        if self._packed is None or self._packed[0] != ver:
            sd = {f"entropy_bottleneck.{k}": v for k, v in self._cpu_params().items()}
            dev = self.quantiles.device
            self._packed = (ver, self.quantiles.detach()[:, 0, 1].contiguous().float().to(dev),
                            eb_pack_params(sd).to(dev))
        return self._packed[1], self._packed[2]

    @staticmethod
    def _build_indexes(size):
        """entropy_models.py:512-523: the table index of a symbol is its channel."""
        C = size[1]
        n = int(np.prod(size[2:]))
        return np.repeat(np.arange(C, dtype=np.int32), n)  # one batch item, flattened


class GaussianConditional(EntropyModel):
    """entropy_models.py:545-685."""

    @staticmethod
    def _validate_scale_table(scale_table):
        """None, or a non-empty ascending list / tuple of positive scales (errors of
        entropy_models.py:556-569)."""
        if scale_table is None:
            return
        if not isinstance(scale_table, (list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        ascending = all(a <= b for a, b in zip(scale_table, scale_table[1:]))
        if not ascending or min(scale_table) <= 0:
            raise ValueError(f'Invalid scale_table "({scale_table})"')

    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self._validate_scale_table(scale_table)
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = scale_table[0]     # entropy_models.py:573-574
        if scale_bound is None or scale_bound <= 0:
            raise ValueError("Invalid parameters")
        self.lower_bound_scale = _LowerBound(scale_bound)
        table = self._prepare_scale_table(scale_table) if scale_table else torch.Tensor()
        self.register_buffer("scale_table", table)
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(inputs):
        return 0.5 * torch.erfc(float(-(2 ** -0.5)) * inputs)

    def update_scale_table(self, scale_table, force=False):
        """entropy_models.py:608-617."""
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        """entropy_models.py:619-643, on the host in float32 (+ scipy float64 ppf)."""
        import scipy.stats
        table = self.scale_table.detach().cpu().float()
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        dev = self.scale_table.device
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._host_tables = None
