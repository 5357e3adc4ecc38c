"""Repeat bench.py's timed region inside one process: spread of the K-frame round-trip time.
    python tools/region_repeat.py --steps 20 --reps 8 --inflight 8 [--pool 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd.pipeline import FramePipeline  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--inflight", type=int, nargs="+", default=[8])
ap.add_argument("--pool", type=int, default=8)
ap.add_argument("--slots", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to(dev)
net.gpu_exclusive = False
net.gpu_slots = a.slots
g = torch.Generator(device=dev)
frames = []
for i in range(a.pool):
    g.manual_seed(1000 + i)
    frames.append(torch.randn((1, 268, 721, 1440), generator=g, device=dev))


def round_trip(x):
    out = net.compress(x)
    x_hat = net.decompress(out["strings"], out["z_shape"])["x_hat"]
    return out, torch.isfinite(x_hat[0, 0, ::97, ::97]).all()


for w in a.inflight:
    pipe = FramePipeline(net, workers=w, device=dev)
    pipe.map(round_trip, [frames[i % a.pool] for i in range(2 * w)])
    ts = []
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.map(round_trip, [frames[i % a.pool] for i in range(a.steps)])
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    fps = [a.steps / t for t in ts]
    print(f"inflight {w:2d} steps {a.steps}: fps " + " ".join(f"{f:5.2f}" for f in fps) +
          f" | median {sorted(fps)[len(fps) // 2]:.2f}")
    pipe.close()
