"""Timeline of GPU / host phases of the frame pipeline (GPU box):
   python tools/phase_timeline.py [--inflight 6] [--frames 18] [--exclusive]"""
import argparse, collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import synth, pipeline
from cra5_amd.zoo import vaeformer_pretrained

ap = argparse.ArgumentParser()
ap.add_argument("--inflight", type=int, default=6)
ap.add_argument("--frames", type=int, default=18)
ap.add_argument("--exclusive", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to(dev)
net.gpu_exclusive = a.exclusive
frames = [synth.synth_frame(268, seed=s).unsqueeze(0).to(dev) for s in (2, 3)]
pipe = pipeline.FramePipeline(net, workers=a.inflight)
pipe.roundtrip([frames[i % 2] for i in range(a.inflight)])   # warm-up
torch.cuda.synchronize()
net.phase_log = log = []
t0 = time.perf_counter()
pipe.roundtrip([frames[i % 2] for i in range(a.frames)])
torch.cuda.synchronize()
t1 = time.perf_counter()
net.phase_log = None
print(f"{a.frames} frames, {a.inflight} in flight, exclusive={a.exclusive}: {(t1-t0)*1e3/a.frames:.1f} ms/frame, {a.frames/(t1-t0):.2f} fps")
by_thread = collections.defaultdict(list)
for th, rq, st, en in log:
    by_thread[th].append((rq, st, en))
gpu, wait, host = [], [], []
names = ["G1 encode", "G2 h_s", "G3 decode"]
per_phase = collections.defaultdict(list)
per_host = collections.defaultdict(list)
for th, ph in by_thread.items():
    ph.sort()
    for i, (rq, st, en) in enumerate(ph):
        k = i % 3
        per_phase[k].append((en - st) * 1e3)
        wait.append((st - rq) * 1e3)
        if i + 1 < len(ph):
            per_host[k].append((ph[i + 1][0] - en) * 1e3)
for k in range(3):
    v = per_phase[k]; h = per_host[k]
    print(f"  {names[k]:10s}: GPU phase mean {sum(v)/len(v):7.2f} ms (min {min(v):6.2f} max {max(v):6.2f});  host time after it mean "
          f"{(sum(h)/len(h) if h else 0):7.2f} ms (min {(min(h) if h else 0):6.2f} max {(max(h) if h else 0):6.2f})")
print(f"  lock wait mean {sum(wait)/len(wait):.2f} ms, max {max(wait):.2f}")
# GPU occupancy from the phase intervals (union)
iv = sorted((st, en) for th, rq, st, en in log)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"  union of GPU phases {busy*1e3:.1f} ms of {(t1-t0)*1e3:.1f} ms wall = {100*busy/(t1-t0):.1f} %; sum of phases {sum(e-s for s,e in iv)*1e3:.1f} ms")
pipe.close()
