"""Budget of ONE serial encode_era5_as_bin + decode_from_bin call pair on a host frame (bench.py single_frame_budget;
VERDICT r5 item 7), under the default and the entropy-matched synthetic weight variants.

  python tools/api_single_frame_budget.py [out.json]
"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from cra5_amd import synth  # noqa: E402
from cra5_amd.api import cra5_api  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402

net = vaeformer_pretrained(268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to("cuda")
api = cra5_api(local_root=tempfile.mkdtemp(), device="cuda", weights=net)
host = (synth.synth_frame(268, seed=1000) * api.std.cpu() + api.mean.cpu()).numpy()
out = {}
for variant in ("default", "matched"):
    synth.apply_variant(net, seed=7, variant=variant)
    b = bench.single_frame_budget(api, host, reps=5)
    out[variant] = b
    print(f"--- weights variant {variant}: {b['frames_per_s']:.2f} frames/s serial")
    for k, v in b.items():
        if k != "frames_per_s":
            print(f"  {k:22s} {v:8.2f} ms")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
