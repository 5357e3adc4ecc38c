"""Reduced-precision attention (hi_only split rows / plain f16 rows): time per launch on the model's shapes and the error
against float64 on the f16-rounded operands (GPU box).   python tools/attn_f16_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, W, C, heads = 72, 144, 1024, 16
N = H * W
g = torch.Generator().manual_seed(1)
qkv = torch.randn(N, 3 * C, generator=g).to(dev)
bias = torch.randn(3 * C, generator=g).to(dev)
qs, ps = ops.split_f16(qkv), ops.split_f16(bias.reshape(1, -1))


def plain_of(sm):
    p = ops.SplitMat.empty(sm.rows, sm.K, dev, zero=True)
    p.data[:, : sm.Kp] = sm.data.view(sm.rows, -1, 2, 32)[:, :, 0].reshape(sm.rows, -1)
    p.plain = True
    return p


qp, pp = plain_of(qs), plain_of(ps)
ok, nb = ops.attention_balanced_plan(N, heads)
ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, (wh, ww) in (("global", (H, W)), ("w24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
    glob = name == "global"
    for lay, (a, p) in (("split", (qs, ps)), ("plain", (qp, pp))):
        o = ops.SplitMat.empty(N, C, dev, zero=True)
        kw = dict(out_split=o, hi_only=True)
        if glob:
            kw.update(workspace=ws, balanced=True)
        dt = timed(lambda: ops.window_attention_split(a, p, heads, H, W, wh, ww, **kw), 5 if glob else 20)
        print(f"hi_only {lay:5s} {name:7s}: {dt * 1e6:8.1f} us  {4.0 * N * wh * ww * C / dt / 1e12:7.1f} TF", flush=True)

# accuracy on one head, window 24 x 24 and a 2304-token global case, against float64 on the f16-rounded q / k / v
for (h_, w_, wh, ww) in ((24, 24, 24, 24), (32, 72, 32, 72)):
    n = h_ * w_
    x = torch.randn(n, 3 * 64, generator=g)
    x16 = x.half().double()
    q, k, v = x16[:, :64], x16[:, 64:128], x16[:, 128:]
    ref = torch.softmax((q * 64 ** -0.5) @ k.t(), -1) @ v
    xs = ops.split_f16(x.to(dev))
    pad = ops.split_f16(torch.zeros(1, 192, device=dev))
    out = ops.window_attention_split(xs, pad, 1, h_, w_, wh, ww, out=torch.empty(n, 64, device=dev), hi_only=True)
    e = float(torch.sqrt(torch.mean((out.double().cpu() - ref) ** 2)))
    r = float(torch.sqrt(torch.mean(ref ** 2)))
    print(f"hi_only accuracy {h_}x{w_}: rmse {e:.3e}  rms(out) {r:.3e}  rel {e / r:.3e}")
