"""Feasibility probe: hipGraph replay of the g_s decode phase (~200 launches) vs eager ctypes launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402

dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False)
synth.load_synthetic(net, seed=7)
net = net.to(dev)
yh = torch.randn(256, 72, 144, device=dev)
for _ in range(2):
    ref = net._decode_frame(yh)
torch.cuda.synchronize()


def timed(fn, n=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (t1 - t0) / n * 1e3


print("eager : device %.2f ms, host issue %.2f ms per frame" % timed(lambda: net._decode_frame(yh)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    net._decode_frame(yh)   # workspaces of this stream/thread exist
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        out = net._decode_frame(yh)
    s.synchronize()
    g.replay()
    s.synchronize()
    print("graph == eager:", torch.equal(out, ref))
    print("graph : device %.2f ms, host issue %.2f ms per frame" % timed(lambda: g.replay()))
