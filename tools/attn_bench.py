"""Micro-benchmark of the attention kernels (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import ops
dev = torch.device("cuda:0")
H, W, C, heads = 72, 144, 1024, 16
qkv = torch.randn(H * W, 3 * C, device=dev)
bias = torch.randn(3 * C, device=dev)
qs = ops.split_f16(qkv); ps = ops.split_f16(bias.reshape(1, -1))
out_s = ops.SplitMat.empty(H * W, C, dev, zero=True)
for name, (wh, ww) in (("global", (72, 144)), ("w24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
    for _ in range(2): ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out_split=out_s)
    torch.cuda.synchronize(); n = 5 if name == "global" else 20
    t0 = time.perf_counter()
    for _ in range(n): ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out_split=out_s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"split {name:8s}: {dt*1e3:8.3f} ms  {4.0*H*W*wh*ww*C/dt/1e12:7.1f} TF", flush=True)

ok, nb = ops.attention_balanced_plan(H * W, heads)
if ok:
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    for _ in range(2): ops.window_attention_split(qs, ps, heads, H, W, H, W, out_split=out_s, workspace=ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.window_attention_split(qs, ps, heads, H, W, H, W, out_split=out_s, workspace=ws)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"split global balanced: {dt*1e3:8.3f} ms  {4.0*H*W*H*W*C/dt/1e12:7.1f} TF", flush=True)

# hyper-prior shape: 18 x 36 tokens, 360-d, 5 heads x 72 (exact-f32 kernel)
H2, W2, C2, h2 = 18, 36, 360, 5
qkv2 = torch.randn(H2 * W2, 3 * C2, device=dev); b2 = torch.randn(3 * C2, device=dev)
o2 = torch.empty(H2 * W2, C2, device=dev)
for _ in range(3): ops.window_attention(qkv2, b2, h2, H2, W2, H2, W2, out=o2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): ops.window_attention(qkv2, b2, h2, H2, W2, H2, W2, out=o2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"f32 hyper 648x72 : {dt*1e6:8.1f} us (CRA5_ATT72_NW={os.environ.get('CRA5_ATT72_NW','auto')})", flush=True)
