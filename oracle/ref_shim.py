"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

Makes the *reference's own Python* (``/root/reference``, read-only, this
container only) importable so that golden vectors can be generated from it
(``tests/golden/make_golden.py``).  Nothing here travels to the GPU box: the
fixtures it produces do.

What is shimmed:
  * third-party modules missing from the image that the reference merely imports
    (timm.models.layers helpers, dict_recursive_update, pytorch_msssim,
    torchvision.transforms) -> tiny in-memory stand-ins for *imports*, not for
    anything on the hot path;
  * ``compressai.*``  ->  alias of the vendored ``cra5.models.compressai.*``;
  * ``compressai._CXX`` -> ``oracle/_ref/_CXX*.so`` compiled from the reference's
    own ``cpp_exts/ops/ops.cpp`` (see oracle/Makefile);
  * ``compressai.ans``  -> NOT buildable (rans64.h is an un-vendored submodule):
    bound to ``oracle/rans_py.py`` classes (our restatement) so that the
    reference's compress()/decompress() *python* code runs end to end.  Byte
    parity with upstream compressai.ans is therefore "parity unpinned".
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import glob
import os
import sys
import types

REF_ROOT = os.environ.get("CRA5_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolve ``compressai[.x]`` to ``cra5.models.compressai[.x]``."""

    PREFIX = "compressai"
    TARGET = "cra5.models.compressai"

    def find_spec(self, fullname, path=None, target=None):
        if fullname == self.PREFIX or fullname.startswith(self.PREFIX + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        real = self.TARGET + spec.name[len(self.PREFIX):]
        mod = importlib.import_module(real)
        return mod

    def exec_module(self, module):
        pass


def install(ans_module=None):
    """Install the shims; returns the reference `cra5` package."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected in the build container only)")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import torch

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def drop_path(x, drop_prob=0.0, training=False):
        assert not training or not drop_prob
        return x

    class DropPath(torch.nn.Identity):
        def __init__(self, drop_prob=None):
            super().__init__()

    layers = dict(drop_path=drop_path, DropPath=DropPath, to_2tuple=to_2tuple,
                  trunc_normal_=torch.nn.init.trunc_normal_)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", **layers)
    _stub("timm.layers", **layers)

    def recursive_update(dst, src):
        for k, v in (src or {}).items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                recursive_update(dst[k], v)
            else:
                dst[k] = v
        return dst

    _stub("dict_recursive_update", recursive_update=recursive_update)
    _stub("pytorch_msssim", ms_ssim=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError()))

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms")
    tvt.__getattr__ = lambda name: _Dummy
    tvt.__all__ = []
    tv.transforms = tvt

    # compiled from the reference's own ops.cpp by oracle/Makefile
    cands = glob.glob(os.path.join(_HERE, "_ref", "_CXX*.so"))
    if not cands:
        raise RuntimeError("oracle/_ref/_CXX*.so missing: run `make -C oracle ref`")
    spec = importlib.util.spec_from_file_location("_CXX", cands[0])
    cxx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cxx)
    sys.modules["cra5.models.compressai._CXX"] = cxx
    sys.modules["compressai._CXX"] = cxx

    if ans_module is None:
        from oracle import rans_py as ans_module
    sys.modules["cra5.models.compressai.ans"] = ans_module
    sys.modules["compressai.ans"] = ans_module

    sys.meta_path.insert(0, _AliasFinder())
    import cra5.models.compressai.zoo  # noqa: F401  (must be entered first)
    import cra5
    return cra5
