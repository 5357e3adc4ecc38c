"""Independent yardstick for the attention kernels: what does somebody else's flash attention issue on
the model's two shapes, on THIS box, in THIS process, with the shader clock read beside it?

MEASUREMENT TOOL ONLY: the vendor kernels are reached through torch.nn.functional.scaled_dot_product_attention on plain
f16 tensors (every backend the image ships is tried: flash, memory-efficient, math) and are never linked into
libcra5_amd.so.  Legs, back to back:

  vendor   SDPA, f16 q / k / v [B, 16, L, 64]: global = [1, 16, 10368, 64], windows = [18, 16, 576, 64]
  product  cra5_window_attention_split: reduced-precision form on plain f16 rows (1 MFMA per product), reduced-precision
           form on split rows, fp32-accurate form (3 MFMAs per product)

Compared are the ISSUED f16 MFMA rates (4 * N * L * C flop per launch, x3 for the fp32-accurate form) and the
clock-independent "issued MFMA flop per shader cycle" (chip peak: 256 CUs x 4 SIMDs x 1024).

  python tools/vendor_attention_yardstick.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cra5_amd import ops  # noqa: E402
from tools.vendor_gemm_yardstick import measure  # noqa: E402

dev = torch.device("cuda:0")
H, W, C, heads = 72, 144, 1024, 16
N = H * W


def plain_of(sm):
    p = ops.SplitMat.empty(sm.rows, sm.K, dev, zero=True)
    p.data[:, : sm.Kp] = sm.data.view(sm.rows, -1, 2, 32)[:, :, 0].reshape(sm.rows, -1)
    p.plain = True
    return p


def window_view(t, wh, ww):
    """[N, 16, 64] token-major -> [windows, 16, wh*ww, 64] (what the reference's window_partition feeds SDPA)."""
    x = t.view(H // wh, wh, W // ww, ww, heads, 64).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(-1, heads, wh * ww, 64).contiguous()


def main():
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(N, 3 * C, generator=g).to(dev)
    bias = torch.randn(3 * C, generator=g).to(dev)
    qs, ps = ops.split_f16(qkv), ops.split_f16(bias.reshape(1, -1))
    qp, pp = plain_of(qs), plain_of(ps)
    _, nb = ops.attention_balanced_plan(N, heads)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    q16, k16, v16 = (qkv[:, i * C:(i + 1) * C].half().view(N, heads, 64) for i in range(3))
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "shapes": {}}
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        backends = [("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION),
                    ("math", SDPBackend.MATH)]
    except ImportError:
        backends, sdpa_kernel = [], None

    for name, (wh, ww) in (("global", (H, W)), ("w24x24", (24, 24))):
        L = wh * ww
        flop = 4.0 * N * L * C
        q, k, v = (window_view(t, wh, ww) for t in (q16, k16, v16))
        res = {}
        for bname, be in backends:
            if bname == "math" and name == "global":
                continue                          # materialises 16 x 10368^2 scores: not a flash kernel, 3.4 GB per launch
            try:
                def fn(be=be):
                    with sdpa_kernel(be):
                        return F.scaled_dot_product_attention(q, k, v)
                fn()
                torch.cuda.synchronize()
                res["vendor_sdpa_" + bname] = measure(fn, flop, target_ms=300.0)
            except Exception as e:  # noqa: BLE001
                res["vendor_sdpa_" + bname] = {"error": repr(e)[:200]}
        glob = name == "global"
        o = ops.SplitMat.empty(N, C, dev, zero=True)
        kw = dict(workspace=ws, balanced=True) if glob else {}
        legs = {
            "product_f16_plain_rows": (lambda: ops.window_attention_split(qp, pp, heads, H, W, wh, ww, out_split=o, hi_only=True, **kw), flop),
            "product_f16_split_rows": (lambda: ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out_split=o, hi_only=True, **kw), flop),
            "product_fp32_accurate": (lambda: ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out_split=o, **kw), 3 * flop),
        }
        for leg, (fn, fl) in legs.items():
            res[leg] = measure(fn, fl, target_ms=300.0)
        for leg, r in res.items():
            if "error" in r:
                print(f"{name:7s} {leg:28s} unavailable: {r['error']}", flush=True)
                continue
            print(f"{name:7s} {leg:28s} {r['us_per_launch']:8.1f} us  {r['tflops_issued']:7.1f} TF f16 issued  "
                  f"clock {r['shader_ghz']:.2f} GHz ({r['shader_ghz_p10']:.2f}-{r['shader_ghz_p90']:.2f})  "
                  f"MFMA flop/clk {100 * r['flop_per_clk_frac']:.1f} % of peak", flush=True)
        out["shapes"][name] = dict(N=N, L=L, heads=heads, head_dim=64, algorithmic_flop=flop, **res)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
