"""GDN / IGDN layer on the HIP kernel `cra5_gdn_f32` - mirrors
cra5/models/compressai/layers/gdn.py:41-92 and ops/parametrizers.py:38-64 of the reference
(same parameters / buffers, so a CompressAI CNN-codec checkpoint's GDN entries load).  Not
executed by VAEformer (the reference imports it unused, vaeformer.py:39); north_star names it."""
import torch
import torch.nn as nn

from . import ops
from .entropy import _LowerBound


class NonNegativeParametrizer(nn.Module):
    """ops/parametrizers.py:38-64."""

    def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = _LowerBound((self.minimum + self.reparam_offset ** 2) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal


class GDN(nn.Module):
    """y[i] = x[i] * rsqrt(beta[i] + sum_j gamma[i, j] x[j]^2)   (inverse: * sqrt)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("cra5_amd.layers.GDN computes only on an MI355X (no CPU fallback)")
        beta = self.beta_reparam(self.beta).contiguous()
        gamma = self.gamma_reparam(self.gamma).contiguous()
        return ops.gdn(x.contiguous(), beta, gamma, inverse=self.inverse)
