"""Rate estimate from the entropy models' likelihoods.

`estimated_bits(out)` / `estimated_bpp(out, num_pixels)` follow the rate term of the reference's
RateDistortionLoss (losses/rate_distortion.py:71-74): sum over the likelihood tensors of
log(likelihood) / (-ln 2 * num_pixels), without the training-time `bpp_weight` factor.  Applied to
`VAEformer.forward(x)["likelihoods"]` it predicts the size of the rANS streams `compress(x)` writes
(with trained weights the coder's overhead over the model entropy is ~1 %; residuals outside the
CDF tables are charged the likelihood floor of 1e-9 = 30 bits by the estimate but cost less as escapes).
"""
import math

import torch


def estimated_bits(out):
    """Total model entropy in bits of a forward() output dict (or of its "likelihoods" dict)."""
    lik = out["likelihoods"] if "likelihoods" in out else out
    return float(sum(torch.log(v.double()).sum() for v in lik.values()) / -math.log(2.0))


def estimated_bpp(out, num_pixels):
    """Bits per pixel, num_pixels = N * H * W of the input frames (rate_distortion.py:66, 71-74)."""
    return estimated_bits(out) / float(num_pixels)
