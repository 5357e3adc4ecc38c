"""Soak run (GPU box): `python tools/soak.py [frames=1500]` - the bench pipeline (12 frames in flight) over many frames of a
small pool of distinct tensors: every re-coded tensor must reproduce its streams (sizes, CRC, escape count), device memory
and the process's RSS must not grow after the first batches.  Prints one JSON line."""
import json
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import dist as D, synth  # noqa: E402
from cra5_amd.pipeline import FramePipeline  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to(dev)
net.gpu_exclusive = False
pool = []
g = torch.Generator(device=dev)
for i in range(8):
    g.manual_seed(2000 + i)
    pool.append(torch.randn((1, 268, 721, 1440), generator=g, device=dev))
pipe = FramePipeline(net, workers=12, device=dev)


def rt(x):
    out = net.compress(x)
    xh = net.decompress(out["strings"], out["z_shape"])["x_hat"]
    return D.frame_stats(0, out["strings"], net.last_n_escape()[0])[1:], bool(torch.isfinite(xh[0, 0, ::97, ::97]).all())


seen, mem, rss, t0, done = {}, [], [], time.time(), 0
batch = 96
while done < n_frames:
    res = pipe.map(rt, [pool[(done + i) % len(pool)] for i in range(batch)])
    for i, (row, ok) in enumerate(res):
        assert ok
        k = (done + i) % len(pool)
        assert seen.setdefault(k, row) == row, f"tensor {k} coded differently at frame {done + i}"
    done += batch
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_reserved(dev) / 2 ** 30)
    rss.append(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20)
el = time.time() - t0
print(json.dumps({"frames": done, "frames_per_s": done / el, "distinct_tensors": len(pool),
                  "streams_reproduced": done - len(pool), "device_reserved_GiB_first_last": [mem[0], mem[-1]],
                  "device_reserved_growth_after_2nd_batch_GiB": mem[-1] - mem[min(1, len(mem) - 1)],
                  "host_maxrss_GiB_first_last": [rss[0], rss[-1]], "range_fallbacks": net.range_fallbacks}))
pipe.close()
