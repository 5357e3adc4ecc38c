"""HBM-bound kernels: achieved GB/s (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import ops
dev = torch.device("cuda:0")
C, H, W = 268, 721, 1440
x = torch.randn(C, H, W, device=dev); mean = torch.randn(C, device=dev); std = torch.rand(C, device=dev) + 0.5
K = C * 110
sm = ops.SplitMat.empty(72 * 144, K, dev, zero=True)
cols = torch.randn(72 * 144, K, device=dev)
out = torch.empty(C, H, W, device=dev)
t = torch.randn(10368, 1024, device=dev); g = torch.randn(1024, device=dev); b = torch.randn(1024, device=dev)
hs = ops.SplitMat.empty(10368, 1024, dev)
def bench(name, fn, nbytes, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:22s} {dt*1e6:8.1f} us  {nbytes/dt/1e9:8.1f} GB/s  ({100*nbytes/dt/8e12:.1f}% of 8 TB/s)", flush=True)
bench("im2col+norm (split)", lambda: ops.im2col(x, 11, 10, 10, 10, mean=mean, std=std, out_split=sm), x.numel()*4 + 10368*sm.Kp*4)
bench("col2im+denorm", lambda: ops.col2im(cols, C, 11, 10, 10, 10, 72, 144, mean=mean, std=std, out=out), cols.numel()*4 + out.numel()*4)
bench("layernorm (split out)", lambda: ops.layernorm(t, g, b, out_split=hs, want_f32=False), 2 * t.numel() * 4, n=50)
