"""The real-checkpoint route end to end on the GPU (zoo/image.py:275-300, zoo/pretrained.py:36-64,
vaeformer.py:168-185, models/base.py:69-89 in the reference):

    CRA5_WEIGHTS=<.pth> [CRA5_LIB=<rangecheck flavour>] python tools/checkpoint_route_probe.py [--api]

loads the checkpoint through `vaeformer_pretrained(268, pretrained=True)` (or, with --api, through the DEFAULT
`cra5_api()` constructor), compresses + decompresses the synthetic frame of seed 2 and prints ONE JSON line: stream
hashes, sizes, finiteness of x_hat, and - under the rangecheck flavour - the split-f16 range counters
(|x| >= 65504 / non-finite events), which is the first thing to look at when a REAL checkpoint is loaded."""
import argparse
import ctypes
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd._lib import lib  # noqa: E402


def range_counts():
    out = (ctypes.c_uint64 * 2)()
    try:
        rc = lib().cra5_debug_range_counts(out, 1)
    except AttributeError:
        return None
    return None if rc else [int(out[0]), int(out[1])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--api", action="store_true")
    ap.add_argument("--seed", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.api:
        from cra5_amd.api import cra5_api
        api = cra5_api(local_root="/tmp/cra5_probe")            # default constructor: pretrained=True via $CRA5_WEIGHTS
        net = api.net
    else:
        from cra5_amd.zoo import vaeformer_pretrained
        net = vaeformer_pretrained(quality=268, pretrained=True).eval().to(dev)
    range_counts()
    x = synth.synth_frame(268, seed=a.seed).unsqueeze(0).to(dev)
    out = net.compress(x)
    rec = net.decompress(out["strings"], out["z_shape"])["x_hat"]
    torch.cuda.synchronize()
    y, z = out["strings"][0][0], out["strings"][1][0]
    print(json.dumps(dict(y_sha=hashlib.sha256(y).hexdigest(), z_sha=hashlib.sha256(z).hexdigest(), y_bytes=len(y),
                          z_bytes=len(z), finite=bool(torch.isfinite(rec).all()), range_counts=range_counts(),
                          gc_cdf_shape=list(net.gaussian_conditional._quantized_cdf.shape),
                          eb_cdf_shape=list(net.entropy_bottleneck._quantized_cdf.shape), via="cra5_api()" if a.api else
                          "vaeformer_pretrained(268, pretrained=True)", lib=os.environ.get("CRA5_LIB", "release"))))


if __name__ == "__main__":
    main()
