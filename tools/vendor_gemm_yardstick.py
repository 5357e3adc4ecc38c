"""Independent yardstick for the "sustained all-CU ceiling" claim.

MEASUREMENT TOOL ONLY: the vendor BLAS is reached through torch.nn.functional.linear on f16 tensors and is never
linked into libcra5_amd.so.  For the model's two K = 1024 shapes and a large square-ish one, on THIS box, back to back:

  vendor   plain-f16 GEMM (f16 in, f16 out, fp32 accumulate), random normal operands
  product  cra5_gemm_nt_split (3 f16 MFMAs per product, fp32 / split-f16 out)

and for each leg the shader clock read by a one-wave sampler kernel that sits on the chip beside the GEMMs
(cra5_clock_sampler_launch / cra5_clock_stamp in the C ABI, csrc/runtime.hip).  Compared are the ISSUED f16 MFMA rates (product: 3 x 2MNK) and the
clock-independent figure "issued MFMA flop per shader cycle" (chip peak: 256 CUs x 4 SIMDs x 1024 = 1 048 576).

  python tools/vendor_gemm_yardstick.py [out.json]
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

from cra5_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
PEAK_FLOP_PER_CLK = 256 * 4 * 32768 / 32.0

SHAPES = [("qkv", 10368, 3072, 1024), ("fc1", 10368, 4096, 1024), ("fc2", 10368, 1024, 4096), ("big", 8192, 2048, 8192)]


def measure(fn, flop_issued, target_ms=400.0):
    """Queue ~target_ms of launches, sample the clock for the middle of it; returns us/launch, TF issued, GHz."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 10
    n = max(20, int(target_ms / per))
    samp = ops.ClockSampler(dev, n_max=256, window_us=500)   # 500 us probes on a side stream
    stamps = torch.zeros(2, dtype=torch.int64, device=dev)
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        samp.n = 0
        e0.record()
        for i in range(n):
            if i == n // 5:                     # the probes join once the chip is busy; stamps bracket the rest
                lib().cra5_clock_stamp(ctypes.c_void_p(stamps.data_ptr()), main)
                for _ in range(min(samp.n_max, int(0.5 * n * per / 0.5))):   # half the queue's length of probes
                    samp.probe()
            fn()
        lib().cra5_clock_stamp(ctypes.c_void_p(stamps.data_ptr() + 8), main)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        w0, w1 = (int(v) for v in stamps.cpu().numpy())
        sm = samp.summary(w0, w1)
        ok = sm is not None and sm["probes"] >= 8
        g = sm["shader_ghz_median"] if ok else float("nan")
        row = dict(us_per_launch=us, tflops_issued=flop_issued / us / 1e6, shader_ghz=g,
                   shader_ghz_p10=sm["shader_ghz_p10"] if ok else float("nan"),
                   shader_ghz_p90=sm["shader_ghz_p90"] if ok else float("nan"),
                   sampler_inside_queue=bool(ok), samples_inside=sm["probes"] if sm else 0, launches=n)
        row["flop_per_clk_frac"] = row["tflops_issued"] * 1e3 / g / PEAK_FLOP_PER_CLK
        if best is None or row["us_per_launch"] < best["us_per_launch"]:
            best = row
    return best


def main():
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__,
           "prefer_hipblaslt": os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"), "shapes": {}}
    g = torch.Generator().manual_seed(0)
    for name, M, N, K in SHAPES:
        a32 = torch.randn(M, K, generator=g).to(dev)
        w32 = (torch.randn(N, K, generator=g) * 0.03).to(dev)
        a16, w16 = a32.half(), w32.half()
        c16 = torch.empty(M, N, dtype=torch.float16, device=dev)
        sa, sw = ops.split_f16(a32), ops.split_f16(w32, "auto")
        c32 = torch.empty(M, N, device=dev)
        osp = ops.SplitMat.empty(M, N, dev, zero=True)
        flop = 2.0 * M * N * K
        legs = {
            "vendor_f16": (lambda: torch.mm(a16, w16.t(), out=c16), flop),
            "product_split_f32out": (lambda: ops.gemm_nt_split(sa, sw, out=c32), 3 * flop),
            "product_split_splitout": (lambda: ops.gemm_nt_split(sa, sw, out_split=osp, want_f32=False), 3 * flop),
        }
        res = {}
        for leg, (fn, fl) in legs.items():
            res[leg] = measure(fn, fl)
            r = res[leg]
            print(f"{name:4s} {M}x{N}x{K} {leg:24s} {r['us_per_launch']:8.1f} us  {r['tflops_issued']:7.1f} TF f16 issued  "
                  f"clock {r['shader_ghz']:.2f} GHz ({r['shader_ghz_p10']:.2f}-{r['shader_ghz_p90']:.2f})  "
                  f"MFMA flop/clk {100 * r['flop_per_clk_frac']:.1f} % of peak  inside={r['sampler_inside_queue']}", flush=True)
        res["product_over_vendor_issued"] = res["product_split_f32out"]["tflops_issued"] / res["vendor_f16"]["tflops_issued"]
        out["shapes"][name] = dict(M=M, N=N, K=K, **res)
        del a32, w32, a16, w16, c16, sa, sw, c32, osp
    # zero-filled operands: how much of the vendor figure is data-dependent clock (guide: +15-21 %)
    M, N, K = 8192, 2048, 8192
    a16 = torch.zeros(M, K, dtype=torch.float16, device=dev)
    w16 = torch.zeros(N, K, dtype=torch.float16, device=dev)
    c16 = torch.empty(M, N, dtype=torch.float16, device=dev)
    r = measure(lambda: torch.mm(a16, w16.t(), out=c16), 2.0 * M * N * K)
    print(f"big  zero-filled vendor_f16 {r['us_per_launch']:8.1f} us {r['tflops_issued']:7.1f} TF clock {r['shader_ghz']:.2f} GHz")
    out["big_zero_filled_vendor"] = r
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
