"""Windowed attention, A / B of the launch forms (GPU box): run once per form, the second run compares with the first.

  python tools/attn_window_ab.py gpurun_out/r06/attn_win                 # 4-wave work-groups (the product path)
  python tools/attn_window_ab.py gpurun_out/r06/attn_win persistent      # persistent 12-wave units (round 6 experiment)

Per window shape of the model (24 x 24, 12 x 48, 48 x 12 on the 72 x 144 grid, 16 heads x 64) and per precision form
(fp32-accurate, reduced precision on split rows, on plain rows): us per launch (HIP events over 200 back-to-back launches)
and, on the second run, the difference to the first run's output (tiles of FULL units run the same arithmetic: identical;
tiles of SPLIT units are merged from two key halves: fp32 noise) and both runs' error against float64 on one window.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
form = "persistent" if len(sys.argv) > 2 and sys.argv[2].startswith("p") else "classic"
prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/attn_win"
H, W, C, heads = 72, 144, 1024, 16
N = H * W
g = torch.Generator().manual_seed(1)
qkv = torch.randn(N, 3 * C, generator=g).to(dev)
qkv[:, : 2 * C] *= 1.5
bias = torch.randn(3 * C, generator=g).to(dev)
qs, ps = ops.split_f16(qkv), ops.split_f16(bias.reshape(1, -1))


def plain_of(sm):
    p = ops.SplitMat.empty(sm.rows, sm.K, dev, zero=True)
    p.data[:, : sm.Kp] = sm.data.view(sm.rows, -1, 2, 32)[:, :, 0].reshape(sm.rows, -1)
    p.plain = True
    return p


qp, pp = plain_of(qs), plain_of(ps)
outs = {}
for name, (wh, ww) in (("w24x24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
    for prec, (a, p, kw) in (("fp32", (qs, ps, {})), ("f16_split", (qs, ps, dict(hi_only=True))),
                             ("f16_plain", (qp, pp, dict(hi_only=True)))):
        o32 = torch.empty(N, C, device=dev) if prec == "fp32" else None
        osp = ops.SplitMat.empty(N, C, dev, zero=True)
        fn = lambda: ops.window_attention_split(a, p, heads, H, W, wh, ww, out=o32, out_split=osp,  # noqa: E731
                                                persistent_units=form == "persistent", **kw)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        res = (o32 if o32 is not None else osp.data.view(torch.int16)).cpu()
        outs[f"{name}_{prec}"] = res
        print(f"{form:10s} {name} {prec:9s}: {best:7.1f} us  {4.0 * N * wh * ww * C / best / 1e6:7.1f} TF"
              f"{'' if prec != 'fp32' else '  (x3 MFMAs issued)'}", flush=True)

# accuracy of the fp32-accurate form against float64, window (0, 0) of each shape, head 3
for name, (wh, ww) in (("w24x24", (24, 24)), ("w12x48", (12, 48)), ("w48x12", (48, 12))):
    rows = (torch.arange(wh)[:, None] * W + torch.arange(ww)[None, :]).reshape(-1)
    rows = rows[(torch.arange(wh)[:, None].expand(wh, ww).reshape(-1) < H)]
    hd = 3
    x = qkv.cpu().double()
    if wh > H:
        continue
    q, k, v = (x[rows][:, i * C + 64 * hd: i * C + 64 * hd + 64] for i in range(3))
    # (48 x 12 windows: window (0, 0) has no padding - rows 0..47 of the 72)
    ref = torch.softmax((q * 64 ** -0.5) @ k.t(), -1) @ v
    got = outs[f"{name}_fp32"][rows][:, 64 * hd: 64 * hd + 64].double()
    e = float(torch.sqrt(torch.mean((got - ref) ** 2)))
    print(f"{form:10s} {name} fp32-accurate vs float64 (window 0, head {hd}): rmse {e:.2e}  rms {float(torch.sqrt(torch.mean(ref ** 2))):.2e}")

other = f"{prefix}_{'persistent' if form == 'classic' else 'classic'}.pt"
torch.save(outs, f"{prefix}_{form}.pt")
if os.path.exists(other):
    ref = torch.load(other)
    for k, v in outs.items():
        if v.dtype == torch.float32:
            d = (v - ref[k]).abs()
            same = float((d == 0).float().mean())
            print(f"{k}: max |persistent - classic| {float(d.max()):.2e}, identical elements {100 * same:.1f} %, finite {bool(torch.isfinite(v).all())}")
        else:
            same = float((v == ref[k]).float().mean())
            print(f"{k}: identical f16 halves {100 * same:.2f} %")
