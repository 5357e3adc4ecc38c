"""cra5_amd - MI355X (gfx950) native VAEformer encode/decode path.

Drop-in for the hot path of taohan10200/CRA5:
    from cra5_amd import cra5_api, vaeformer_pretrained
or, to keep the reference's import lines unchanged, call `install_dropin()` once and
    from cra5.api import cra5_api
    from cra5.models.compressai.zoo import vaeformer_pretrained
"""
import sys
import types

__version__ = "0.1.0"


def __getattr__(name):
    if name == "cra5_api":
        from .api import cra5_api
        return cra5_api
    if name == "vaeformer_pretrained":
        from .zoo import vaeformer_pretrained
        return vaeformer_pretrained
    if name == "VAEformer":
        from .vaeformer import VAEformer
        return VAEformer
    raise AttributeError(name)


def install_dropin():
    """Register `cra5.api` and `cra5.models.compressai.zoo` aliases in sys.modules so code
    written against the reference's import paths runs on this package."""
    from . import api, zoo, vaeformer

    def mod(name, **attrs):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        m.__dict__.update(attrs)
        return m

    mod("cra5")
    mod("cra5.api", cra5_api=api.cra5_api)
    mod("cra5.models")
    mod("cra5.models.compressai")
    mod("cra5.models.compressai.zoo", vaeformer_pretrained=zoo.vaeformer_pretrained,
        bmshj2018_factorized=zoo.bmshj2018_factorized, bmshj2018_factorized_relu=zoo.bmshj2018_factorized_relu,
        bmshj2018_hyperprior=zoo.bmshj2018_hyperprior, mbt2018_mean=zoo.mbt2018_mean)
    mod("cra5.models.vaeformer")
    mod("cra5.models.vaeformer.vaeformer", VAEformer=vaeformer.VAEformer)
