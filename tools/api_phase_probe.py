"""Where does a frame's time go in the PCIe-inclusive batch API (bench.py `api_pipelined`), and what does the link gate buy?

encode_era5_batch / decode_batch / roundtrip_batch of the 268 model on HOST fp32 frames, n frames, 12 in flight, with
cra5_api.phase_log on: per phase (pageable -> pinned memcpy, wait for the link, transfer, compress, decompress, consumer
copy) the mean duration per frame and the fraction of the wall time the link was busy in each direction.  A / B inside one
process: link_serial False (round 5: every frame thread issues its DMA when it gets there) vs True (one frame per direction
at a time).

  python tools/api_phase_probe.py [n_frames=72] [out.json]
"""
import json
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd.api import cra5_api  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 72
net = vaeformer_pretrained(268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to("cuda")
net.gpu_exclusive = False
tmp = tempfile.mkdtemp()
api = cra5_api(local_root=tmp, device="cuda", weights=net)
host = [(synth.synth_frame(268, seed=5 + i) * api.std.cpu() + api.mean.cpu()).numpy() for i in range(6)]
stamps = [f"2024-06-{1 + i // 24:02d}T{i % 24:02d}:00:00" for i in range(n)]
data = [host[i % len(host)] for i in range(n)]
tls = threading.local()


def consumer(i, arr):
    dst = getattr(tls, "dst", None)
    if dst is None:
        dst = tls.dst = np.empty(arr.shape, np.float32)
    np.copyto(dst, arr)
    return 0


def summarise(log, wall, frames):
    out = {}
    for ph in sorted({p for _, p, _, _ in log}):
        d = [t1 - t0 for _, p, t0, t1 in log if p == ph]
        out[ph] = {"mean_ms": 1e3 * sum(d) / len(d), "max_ms": 1e3 * max(d), "per_frame_ms": 1e3 * sum(d) / frames,
                   "busy_frac_of_wall": sum(d) / wall}
    return out


def run(workers, serial, what):
    api.link_serial = serial
    W = workers
    res = {}
    for rep in range(2):                         # rep 0 warms pinned buffers / workspaces
        nn = n if rep else W
        api.phase_log = log = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if what == "encode+decode":
            enc = api.encode_era5_batch(stamps[:nn], data=data[:nn], save_root=tmp + "/E", workers=W)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            log_e = list(log)
            api.phase_log = log = []
            api.decode_batch(paths=[e["save_path"] for e in enc], workers=W, sink=consumer)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res = {"encode_fps": nn / (t1 - t0), "decode_fps": nn / (t2 - t1),
                   "encode_phases": summarise(log_e, t1 - t0, nn), "decode_phases": summarise(log, t2 - t1, nn)}
        else:
            api.roundtrip_batch(stamps[:nn], data=data[:nn], save_root=tmp + "/R", workers=W, sink=consumer)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = {"round_trip_streamed_fps": nn / (t1 - t0), "phases": summarise(log, t1 - t0, nn)}
    api.phase_log = None
    return res


out = {"frames": n, "runs": []}
for workers in (12, 16):
    for serial in (False, True):
        r = {"workers": workers, "link_serial": serial}
        r.update(run(workers, serial, "encode+decode"))
        r.update(run(workers, serial, "roundtrip"))
        out["runs"].append(r)
        print(f"workers {workers} link_serial {serial}: encode {r['encode_fps']:.1f} decode {r['decode_fps']:.1f} streamed round trip "
              f"{r['round_trip_streamed_fps']:.1f} frames/s", flush=True)
        for side in ("encode_phases", "decode_phases", "phases"):
            print("   ", side, {k: f"{v['mean_ms']:.1f} ms (busy {v['busy_frac_of_wall']:.2f})" for k, v in r[side].items()}, flush=True)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
