"""Board power / shader clock while the hot kernels run (GPU box).

Is the chip power-limited under the split-f16 GEMM?  A sampler thread reads the amdgpu hwmon files
(power1_average / power1_input in uW, power1_cap, freq1_input = sclk in Hz) every 20 ms while the main thread
loops one kernel for a few seconds: the qkv / fc2 GEMM, the whole-grid attention, LayerNorm (HBM-bound), idle.

  python tools/power_probe.py [seconds-per-leg]
"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402


def _hwmon():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        names = os.listdir(d)
        if any(n.startswith("power1") for n in names):
            return d
    return None


def _read(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, hw):
        super().__init__(daemon=True)
        self.hw, self.rows, self.on = hw, [], True
        self.pfile = next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))), None)
        self.ffile = os.path.join(hw, "freq1_input")

    def run(self):
        while self.on:
            self.rows.append((time.time(), _read(self.pfile) if self.pfile else None, _read(self.ffile)))
            time.sleep(0.02)


def leg(name, fn, secs, smp):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    t1 = time.time()
    rows = [r for r in smp.rows if t0 + 0.3 * secs <= r[0] <= t1]
    pw = [r[1] for r in rows if r[1]]
    fq = [r[2] for r in rows if r[2]]
    print(f"{name:14s} {1e6 * (t1 - t0) / max(n, 1):9.1f} us/launch   power {sum(pw) / max(len(pw), 1) / 1e6:7.1f} W "
          f"(max {max(pw, default=0) / 1e6:7.1f})   sclk {sum(fq) / max(len(fq), 1) / 1e6:7.1f} MHz (min {min(fq, default=0) / 1e6:.0f} max {max(fq, default=0) / 1e6:.0f})",
          flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    hw = _hwmon()
    print("hwmon:", hw, sorted(os.listdir(hw)) if hw else None)
    if hw:
        print("power cap (W):", (_read(os.path.join(hw, "power1_cap")) or 0) / 1e6, "cap max:", (_read(os.path.join(hw, "power1_cap_max")) or 0) / 1e6)
    os.system("rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -v '^$' | head -30")
    smp = Sampler(hw)
    if hw:
        smp.start()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)

    def gemm(M, N, K, epi):
        a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
        w = ops.split_f16((torch.randn(N, K, generator=g) * 0.03).to(dev), "auto")
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev) if epi == "res" else None
        out = torch.empty(M, N, device=dev) if "split" not in epi else None
        osp = ops.SplitMat.empty(M, N, dev, zero=True) if "split" in epi else None
        kw = dict(bias=b, res=r, gelu="gelu" in epi, out=out, out_split=osp, want_f32=out is not None)
        return lambda: ops.gemm_nt_split(a, w, **kw)

    legs = [("idle", lambda: None), ("qkv gemm", gemm(10368, 3072, 1024, "bias_split")), ("fc2 gemm", gemm(10368, 1024, 4096, "res")),
            ("unembed gemm", gemm(10368, 29480, 1024, "none"))]
    # whole-grid attention on split q/k/v
    try:
        qkv = ops.SplitMat.empty(10368, 3072, dev, zero=True)
        x = torch.randn(10368, 1024, generator=g).to(dev)
        wq = ops.split_f16((torch.randn(3072, 1024, generator=g) * 0.03).to(dev), "auto")
        ops.gemm_nt_split(ops.split_f16(x), wq, out_split=qkv, want_f32=False)
        osp = ops.SplitMat.empty(10368, 1024, dev, zero=True)
        ps = ops.split_f16(torch.randn(1, 3072, generator=g).to(dev))
        nb = ops.attention_workspace_bytes(10368, 16)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        legs.append(("global attn", lambda: ops.window_attention_split(qkv, ps, 16, 72, 144, 72, 144, out_split=osp, workspace=ws)))
        legs.append(("window attn", lambda: ops.window_attention_split(qkv, ps, 16, 72, 144, 24, 24, out_split=osp)))
    except Exception as e:  # signature drift: the probe is about power, not about this leg
        print("attention leg skipped:", e)
    xln = torch.randn(10368, 1024, generator=g).to(dev)
    wln = torch.ones(1024, device=dev)
    try:
        oln = ops.SplitMat.empty(10368, 1024, dev, zero=True)
        legs.append(("layernorm", lambda: ops.layernorm(xln, wln, wln, 1e-6, out_split=oln)))
    except Exception as e:
        print("layernorm leg skipped:", e)
    for name, fn in legs:
        try:
            fn()
            leg(name, fn, secs if name != "idle" else 1.0, smp)
        except Exception as e:
            print(name, "failed:", repr(e)[:200])
    smp.on = False


if __name__ == "__main__":
    main()
