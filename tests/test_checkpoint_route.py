"""The `pretrained=True` route (the first thing a user with the real `cra5_268v_300k.pth` does): a synthetic 268
checkpoint written in the REFERENCE's on-disk layout, loaded through `$CRA5_WEIGHTS` by
`vaeformer_pretrained(268, pretrained=True)` and by the default `cra5_api()` constructor.

Reference: zoo/image.py:275-300 (`_load_model`), zoo/pretrained.py:36-64 (`rename_key` / `load_pretrained`),
vaeformer.py:168-185 (`from_state_dict`: `backbone.` prefix stripped, `kl_loss.logvar` dropped),
models/base.py:69-89 (`load_state_dict` resizes the empty CDF buffers to the checkpoint's)."""
import hashlib
import json
import math
import os
import subprocess
import sys
from collections import OrderedDict

import pytest
import torch

from cra5_amd import synth
from cra5_amd import zoo
from cra5_amd.zoo import vaeformer_pretrained

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from cra5_amd.vaeformer import VAEformer


class ThinVAEformer(VAEformer):
    """`from_state_dict` builds `cls(variable_num)` (vaeformer.py:168-185): the CPU tests register this thin-width
    full-spatial architecture (8 variables) under the zoo's "vaeformer-pretrained" key so that the WHOLE route runs in
    seconds; the 1.6 GB 268 checkpoint goes through the same code in the -m gpu test below."""

    def __init__(self, variable_num):
        assert variable_num == 8
        super().__init__(0, **synth.thin_model_kwargs())


def _source_model(cls=None):
    """Synthetic weights and CDF tables of NON-default size (a 64-entry scale table up to 300 instead of 256: the
    checkpoint's int32 buffers then differ in shape from what update() would build)."""
    net = cls(8) if cls is not None else vaeformer_pretrained(quality=268, pretrained=False)
    synth.load_synthetic(net, seed=7, update=False)
    net.update(scale_table=torch.exp(torch.linspace(math.log(0.11), math.log(300.0), 64)), force=True)
    return net


def _checkpoint_dict(net, layout):
    """state_dict of `net` in the reference's on-disk layouts."""
    sd = OrderedDict()
    for k, v in net.state_dict().items():
        v = v.detach().cpu().clone()
        if layout == "legacy_eb" and k.startswith("entropy_bottleneck._"):
            # pre-1.2 CompressAI: nn.ParameterList names, un-prefixed (rename_key only rewrites keys that START with
            # "entropy_bottleneck.", zoo/pretrained.py:48-56)
            for new, old in (("_matrix", "_matrices."), ("_bias", "_biases."), ("_factor", "_factors.")):
                if k.startswith(f"entropy_bottleneck.{new}") and k[-1].isdigit():
                    sd[f"entropy_bottleneck.{old}{k[-1]}"] = v
                    break
            else:
                sd["backbone." + k] = v
            continue
        sd[("module.backbone." if layout == "module" else "backbone.") + k] = v
    sd[("module." if layout == "module" else "") + "kl_loss.logvar"] = torch.zeros(())     # training-only key
    return {"state_dict": sd, "meta": {"iter": 300000}} if layout == "wrapped" else sd


@pytest.fixture(scope="module")
def source():
    return _source_model(ThinVAEformer)


@pytest.fixture(scope="module")
def ckpt_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("ckpt")


def _assert_same_state(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert sa[k].dtype == sb[k].dtype and sa[k].shape == sb[k].shape and torch.equal(sa[k].cpu(), sb[k].cpu()), k


@pytest.mark.parametrize("layout", ["backbone", "wrapped", "module", "legacy_eb"])
def test_pretrained_route_loads_reference_layouts(source, ckpt_dir, layout, monkeypatch):
    path = str(ckpt_dir / f"cra5_268v_{layout}.pth")
    torch.save(_checkpoint_dict(source, layout), path)
    monkeypatch.setenv("CRA5_WEIGHTS", path)
    monkeypatch.setitem(zoo.model_architectures, "vaeformer-pretrained", ThinVAEformer)
    net = vaeformer_pretrained(quality=268, pretrained=True)
    assert isinstance(net, ThinVAEformer)
    _assert_same_state(source, net)
    gc, eb = net.gaussian_conditional, net.entropy_bottleneck
    assert gc._quantized_cdf.dtype == torch.int32 and gc._quantized_cdf.shape == source.gaussian_conditional._quantized_cdf.shape
    assert gc._quantized_cdf.shape[1] != vaeformer_default_gc_width()      # really a non-default table
    assert eb._quantized_cdf.dtype == torch.int32 and eb._quantized_cdf.numel() > 0
    assert not any("kl_loss" in k for k in net.state_dict())
    os.remove(path)


_DEFAULT_W = []


def vaeformer_default_gc_width():
    if not _DEFAULT_W:
        from cra5_amd.entropy import GaussianConditional, get_scale_table
        gc = GaussianConditional(None)
        gc.update_scale_table(get_scale_table(), force=True)
        _DEFAULT_W.append(gc._quantized_cdf.shape[1])
    return _DEFAULT_W[0]


def test_pretrained_route_errors_match_reference(monkeypatch, tmp_path):
    monkeypatch.delenv("CRA5_WEIGHTS", raising=False)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    with pytest.raises(RuntimeError, match="Pre-trained model not yet available"):
        vaeformer_pretrained(quality=268, pretrained=True)
    # $CRA5_WEIGHTS holds VAEformer checkpoints only: the CNN entries keep the reference's error
    bogus = tmp_path / "x.pth"
    torch.save({"backbone.g_a.patch_embed.proj.weight": torch.zeros(1, 268, 11, 10)}, str(bogus))
    monkeypatch.setenv("CRA5_WEIGHTS", str(bogus))
    for entry in (zoo.bmshj2018_factorized, zoo.bmshj2018_factorized_relu, zoo.bmshj2018_hyperprior, zoo.mbt2018_mean):
        with pytest.raises(RuntimeError, match="Pre-trained model not yet available"):
            entry(1, pretrained=True)
    with pytest.raises(RuntimeError, match="Pre-trained model not yet available"):
        vaeformer_pretrained(quality=268, metric="ms-ssim", pretrained=True)


def test_dropin_zoo_exports_match_reference():
    """zoo/__init__.py of the reference exports the ReLU variant too."""
    import cra5_amd
    cra5_amd.install_dropin()
    import cra5.models.compressai.zoo as z
    for name in ("vaeformer_pretrained", "bmshj2018_factorized", "bmshj2018_factorized_relu", "bmshj2018_hyperprior",
                 "mbt2018_mean"):
        assert callable(getattr(z, name)), name


@pytest.mark.gpu
def test_pretrained_route_end_to_end_on_gpu(ckpt_dir, dev, monkeypatch):
    """Checkpoint on disk -> `vaeformer_pretrained(268, pretrained=True)` AND the default `cra5_api()` constructor
    -> compress: streams equal the in-memory model's byte for byte; the same run under the `rangecheck` flavour
    prints the split-f16 range counters."""
    from cra5_amd import build as B
    source = _source_model()            # the real 268 architecture (404.7 M parameters, 1.6 GB on disk)
    path = str(ckpt_dir / "cra5_268v_300k.pth")
    torch.save(_checkpoint_dict(source, "backbone"), path)
    monkeypatch.setenv("CRA5_WEIGHTS", path)
    x = synth.synth_frame(268, seed=2).unsqueeze(0).to(dev)
    ref = source.to(dev).compress(x)
    want = [hashlib.sha256(ref["strings"][i][0]).hexdigest() for i in (0, 1)]
    source.to("cpu")
    net = vaeformer_pretrained(quality=268, pretrained=True).eval().to(dev)
    out = net.compress(x)
    assert out["strings"][0][0] == ref["strings"][0][0] and out["strings"][1][0] == ref["strings"][1][0]
    del net, out
    torch.cuda.empty_cache()
    env = dict(os.environ, CRA5_WEIGHTS=path, CRA5_LIB=B.build(flavour="rangecheck"))
    for flag in ([], ["--api"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "checkpoint_route_probe.py")] + flag, env=env,
                           cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print("checkpoint route", d)
        assert [d["y_sha"], d["z_sha"]] == want and d["finite"]
        assert d["range_counts"] == [0, 0]          # synthetic weights: nothing leaves the f16 range
        assert d["gc_cdf_shape"] == list(source.gaussian_conditional._quantized_cdf.shape)
    os.remove(path)
