"""Where the time of a windowed attention launch goes, from in-kernel wall-clock stamps (needs the trace variant):
   tools/build_variant.sh atrace attention_split_f16.hip -DCRA5_ATTN_TRACE
   CRA5_LIB=build_variants/libcra5_atrace.so python tools/attn_trace.py

Per work-group (product path: 1440 four-wave work-groups) / per unit (persistent 12-wave units: up to 3 per work-group):
prologue (start -> key loop), key loop, epilogue (-> end), the shader clock, and the launch's span."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402
from cra5_amd._lib import lib  # noqa: E402

L = lib()
L.cra5_debug_attn_trace.restype = ctypes.c_int
L.cra5_debug_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
H, W, C, heads = 72, 144, 1024, 16
N = H * W
g = torch.Generator().manual_seed(1)
qkv = torch.randn(N, 3 * C, generator=g).to(dev)
bias = torch.randn(3 * C, generator=g).to(dev)
qs, ps = ops.split_f16(qkv), ops.split_f16(bias.reshape(1, -1))
out = ops.SplitMat.empty(N, C, dev, zero=True)
buf = np.zeros(6 * 8192, np.uint64)
for form in ("classic", "persistent"):
    for hi in (False, True):
        for _ in range(3):
            ops.window_attention_split(qs, ps, heads, H, W, 24, 24, out_split=out, hi_only=hi, persistent_units=form == "persistent")
        torch.cuda.synchronize()
        L.cra5_debug_attn_trace(buf.ctypes.data, 8192)         # clear
        ops.window_attention_split(qs, ps, heads, H, W, 24, 24, out_split=out, hi_only=hi, persistent_units=form == "persistent")
        L.cra5_debug_attn_trace(buf.ctypes.data, 8192)
        t = buf.reshape(8192, 6).astype(np.int64)
        t = t[t[:, 0] > 0]
        us = lambda a: a / 100.0          # 100 MHz wall clock  # noqa: E731
        span = us(t[:, 3].max() - t[:, 0].min())
        pro, loop, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
        ghz = t[:, 5] / np.maximum(1, (t[:, 3] - t[:, 0]) * 10.0)
        print(f"--- {form}, {'reduced precision' if hi else 'fp32-accurate'}: {len(t)} rows, launch span {span:.1f} us, clock {np.median(ghz):.2f} GHz")
        dur = us(t[:, 3] - t[:, 0])
        st = us(t[:, 0] - t[:, 0].min())
        q = lambda a, p: float(np.percentile(a, p))  # noqa: E731
        print(f"    duration per row: p5 {q(dur, 5):.1f} p50 {q(dur, 50):.1f} p95 {q(dur, 95):.1f} max {dur.max():.1f} us; "
              f"rows starting in the first 2 us: {int((st < 2).sum())}, duration of those p50 {q(dur[st < 2], 50):.1f} p95 {q(dur[st < 2], 95):.1f} "
              f"max {dur[st < 2].max():.1f}; later rows: start p5 {q(st[st >= 2], 5) if (st >= 2).any() else 0:.1f} p50 "
              f"{q(st[st >= 2], 50) if (st >= 2).any() else 0:.1f} p95 {q(st[st >= 2], 95) if (st >= 2).any() else 0:.1f}, duration p50 "
              f"{q(dur[st >= 2], 50) if (st >= 2).any() else 0:.1f} p95 {q(dur[st >= 2], 95) if (st >= 2).any() else 0:.1f}")
        for nm, m in (("first round (start < 2 us)", st < 2), ("later rows", st >= 2)):
            if m.any():
                print(f"    {nm}: prologue p50 {q(pro[m], 50):.2f} us, key loop p50 {q(loop[m], 50):.2f} = {q(loop[m], 50) / max(1, np.median(t[m, 4] % 1000)):.2f} us / step, "
                      f"epilogue p50 {q(epi[m], 50):.2f}")
        for kind in sorted(set(t[:, 4])):
            m = t[:, 4] == kind
            nt = kind % 1000
            print(f"    {'SPLIT' if kind >= 1000 else 'FULL '} {nt:2d} key tiles x {int(m.sum()):4d}: prologue {np.median(pro[m]):5.2f} us, key loop {np.median(loop[m]):6.2f} us "
                  f"= {np.median(loop[m]) / nt:5.2f} us / step, epilogue {np.median(epi[m]):5.2f} us; start {us(t[m, 0].min() - t[:, 0].min()):.1f}-{us(t[m, 0].max() - t[:, 0].min()):.1f} us, "
                  f"end {us(t[m, 3].min() - t[:, 0].min()):.1f}-{us(t[m, 3].max() - t[:, 0].min()):.1f} us")
