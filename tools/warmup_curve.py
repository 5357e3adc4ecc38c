"""Per-batch frame time of the pipeline from a cold start (GPU box): how long does the slow start last?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import synth, pipeline
from cra5_amd.zoo import vaeformer_pretrained
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False); synth.load_synthetic(net, seed=7); net = net.to(dev)
net.gpu_exclusive = False
frames = [synth.synth_frame(268, seed=s).unsqueeze(0).to(dev) for s in (2, 3)]
pipe = pipeline.FramePipeline(net, workers=8)
t_start = time.perf_counter()
for b in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    t0 = time.perf_counter()
    pipe.roundtrip([frames[i % 2] for i in range(16)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {b:2d} at t={t0 - t_start:6.1f}s: {dt*1e3/16:6.1f} ms/frame ({16/dt:5.1f} fps)", flush=True)
pipe.close()
