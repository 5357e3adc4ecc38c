"""`vaeformer_pretrained` entry point - mirrors
cra5/models/compressai/zoo/image.py:302-324 (+ `_load_model` :275-300, `load_pretrained` /
`rename_key` zoo/pretrained.py:36-64) of the reference.

Differences, stated: quality 159 is accepted (same architecture, 159 variables; the
reference's zoo raises ValueError for anything but 268); `pretrained=True` cannot download
(no network) - the checkpoint is read from `$CRA5_WEIGHTS` or
`~/.cache/torch/hub/checkpoints/cra5_268v_300k.pth` if present, otherwise the reference's
RuntimeError("Pre-trained model not yet available") is raised.
"""
import os

import torch

from .vaeformer import VAEformer

from . import cnn as _cnn

__all__ = ["vaeformer_pretrained", "bmshj2018_factorized", "bmshj2018_factorized_relu", "bmshj2018_hyperprior", "mbt2018_mean",
           "load_pretrained",
           "rename_key", "model_architectures", "cfgs"]

# zoo/image.py:56-62 / :202-245 (the autoregressive `mbt2018` and the cheng2020 models are not built; the reference
# registers the ReLU variant's class under an underscore key its own `_load_model` never finds - it is reachable here)
model_architectures = {"vaeformer-pretrained": VAEformer, "bmshj2018-factorized": _cnn.FactorizedPrior,
                       "bmshj2018-factorized-relu": _cnn.FactorizedPriorReLU,
                       "bmshj2018-hyperprior": _cnn.ScaleHyperprior, "mbt2018-mean": _cnn.MeanScaleHyperprior}
cfgs = dict({"vaeformer-pretrained": {268: (268,), 159: (159,)}}, **_cnn.CNN_CFGS)
_CKPT_NAMES = {268: "cra5_268v_300k.pth"}  # zoo/image.py:69-75


def rename_key(key):
    """zoo/pretrained.py:36-58."""
    if key.startswith("module."):
        key = key[7:]
    if ".downsample." in key:
        return key.replace("downsample", "skip")
    if key.startswith("entropy_bottleneck."):
        if key.startswith("entropy_bottleneck._biases."):
            return f"entropy_bottleneck._bias{key[-1]}"
        if key.startswith("entropy_bottleneck._matrices."):
            return f"entropy_bottleneck._matrix{key[-1]}"
        if key.startswith("entropy_bottleneck._factors."):
            return f"entropy_bottleneck._factor{key[-1]}"
    return key


def load_pretrained(state_dict):
    """zoo/pretrained.py:61-64."""
    return {rename_key(k): v for k, v in state_dict.items()}


def _find_checkpoint(architecture, quality):
    """$CRA5_WEIGHTS / the hub cache hold VAEformer checkpoints only: the CNN architectures have no published
    weights in the reference either (zoo/image.py:290 raises for them)."""
    if architecture != "vaeformer-pretrained":
        return None
    from .config import RuntimeConfig
    cands = [RuntimeConfig.from_env().weights or None]       # CRA5_WEIGHTS (cra5_amd/config.py)
    name = _CKPT_NAMES.get(quality)
    if name:
        cands.append(os.path.join(torch.hub.get_dir(), "checkpoints", name))
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


def _load_model(architecture, metric, quality, pretrained=False, progress=True, **kwargs):
    if architecture not in model_architectures:
        raise ValueError(f'Invalid architecture name "{architecture}"')
    if quality not in cfgs[architecture]:
        raise ValueError(f'Invalid quality value "{quality}"')
    if pretrained:
        path = _find_checkpoint(architecture, quality) if metric == "mse" else None
        if path is None:
            raise RuntimeError("Pre-trained model not yet available")
        state_dict = torch.load(path, map_location="cpu")
        if "state_dict" in state_dict:
            state_dict = state_dict["state_dict"]
        return model_architectures[architecture].from_state_dict(load_pretrained(state_dict))
    return model_architectures[architecture](*cfgs[architecture][quality], **kwargs)


def vaeformer_pretrained(quality, metric="mse", pretrained=False, progress=True, **kwargs):
    """zoo/image.py:302-324."""
    if metric not in ("mse", "ms-ssim"):
        raise ValueError(f'Invalid metric "{metric}"')
    if quality < 1 or quality > 999:
        raise ValueError(f'Invalid quality "{quality}", should be between (1, 999)')
    return _load_model("vaeformer-pretrained", metric, quality, pretrained, progress, **kwargs)


def _cnn_entry(architecture):
    def entry(quality, metric="mse", pretrained=False, progress=True, **kwargs):
        if metric not in ("mse", "ms-ssim"):
            raise ValueError(f'Invalid metric "{metric}"')
        if quality < 1 or quality > 8:
            raise ValueError(f'Invalid quality "{quality}", should be between (1, 8)')
        return _load_model(architecture, metric, quality, pretrained, progress, **kwargs)
    entry.__doc__ = f"zoo/image.py entry point of `{architecture}` (google.py:64-508) on the HIP kernels (cra5_amd/cnn.py)."
    return entry


bmshj2018_factorized = _cnn_entry("bmshj2018-factorized")     # zoo/image.py:326-348
bmshj2018_factorized_relu = _cnn_entry("bmshj2018-factorized-relu")   # zoo/image.py:351-373
bmshj2018_hyperprior = _cnn_entry("bmshj2018-hyperprior")     # zoo/image.py:376-398
mbt2018_mean = _cnn_entry("mbt2018-mean")                     # zoo/image.py:401-423
