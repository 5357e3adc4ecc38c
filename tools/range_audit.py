"""Range audit of the split-f16 engine (needs the `rangecheck` build flavour):

    CRA5_LIB=cra5_amd/_flavours/libcra5_rangecheck.so python tools/range_audit.py [--model thin|268]

Runs (1) GEMM + LayerNorm + attention on activations scaled by 1e-6, 1, 1e4 and 1e5 and (2) a full
encode -> decode of the model on a synthetic frame, and prints ONE JSON line with, per case, the
number of split-f16 stores that saw |x| >= 65 504 (poisoned to inf / -inf by the split, csrc/split.h: the frame
is then re-run on the exact-f32 engines by the model's range guard) or a non-finite value, and the accuracy against float64.  With a real checkpoint loaded through
cra5_api the same counters answer "do this model's activations stay in range" (VERDICT r1).
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import ops, synth  # noqa: E402
from cra5_amd._lib import lib  # noqa: E402


def counts(reset=True):
    out = (ctypes.c_uint64 * 2)()
    rc = lib().cra5_debug_range_counts(out, int(reset))
    if rc:
        raise SystemExit(f"cra5_debug_range_counts rc={rc}: run with CRA5_LIB=<rangecheck flavour>")
    return int(out[0]), int(out[1])


def rel(a, b):
    b = b.double().cpu()
    return float(torch.sqrt(torch.mean((a.double().cpu() - b) ** 2)) / torch.sqrt(torch.mean(b ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="thin")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    g = torch.Generator().manual_seed(3)
    M, N, K = 2048, 1024, 1024
    a0 = torch.randn(M, K, generator=g)
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    sw = ops.split_f16(w, "auto")
    counts()
    for name, scale in (("1e-6", 1e-6), ("1", 1.0), ("1e4", 1e4), ("1e5", 1e5)):
        x = (a0 * scale).to(dev)
        sa = ops.split_f16(x)
        c_in = counts()
        out_s = ops.SplitMat.empty(M, N, dev, zero=True)
        out = ops.gemm_nt_split(sa, sw, out_split=out_s)
        c_out = counts()
        ref = (a0.double() * scale) @ w.double().cpu().t()
        # LayerNorm is scale-invariant: its split output must not depend on the input scale
        ga, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        sm = ops.SplitMat.empty(M, K, dev)
        ops.layernorm(x, ga, be, 1e-6, out_split=sm, want_f32=False)
        ln = sm.to_float()
        c_ln = counts()
        # rows holding an out-of-range element are poisoned (non-finite) as a whole, every other row is exact
        over = ((a0.double() * scale).abs() >= 65520).any(dim=1).to(dev)
        poisoned = (~torch.isfinite(out)).any(dim=1)
        clean = ~over
        res[name] = dict(split_in=c_in, gemm_out=c_out, layernorm=c_ln,
                         gemm_rel_rmse=rel(out[clean], ref[clean.cpu()]) if bool(clean.any()) else None,
                         rows_over=int(over.sum()), rows_poisoned=int(poisoned.sum()),
                         poisoned_equals_over=bool(torch.equal(poisoned, over)),
                         gemm_finite=bool(torch.isfinite(out).all()), ln_finite=bool(torch.isfinite(ln).all()),
                         ln_rms=float(ln.pow(2).mean().sqrt()))
    # non-finite values must STAY non-finite through every split producer (a NaN that v_med3 turned into -65504
    # would make a blown-up frame look healthy to every isfinite() probe downstream): one NaN and one inf in the
    # activations -> the split copy, the GEMM output row (fp32 and split) and the LayerNorm row are non-finite,
    # every other row stays finite
    xn = a0.clone()
    xn[3, 17] = float("nan")
    xn[900, 5] = float("inf")
    xn = xn.to(dev)
    sa = ops.split_f16(xn)
    back = sa.to_float()
    out_s = ops.SplitMat.empty(M, N, dev, zero=True)
    out = ops.gemm_nt_split(sa, sw, out_split=out_s)
    sm = ops.SplitMat.empty(M, K, dev)
    ops.layernorm(xn, torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-6, out_split=sm, want_f32=False)
    ln = sm.to_float()
    bad_rows = torch.zeros(M, dtype=torch.bool, device=dev)
    bad_rows[3] = bad_rows[900] = True

    def rows_nonfinite(t):
        return (~torch.isfinite(t)).any(dim=1)
    res["nonfinite"] = dict(
        split_keeps=bool(not torch.isfinite(back[3, 17]) and not torch.isfinite(back[900, 5])
                         and int((~torch.isfinite(back)).sum()) == 2),
        gemm_rows=bool(torch.equal(rows_nonfinite(out), bad_rows)),
        gemm_split_rows=bool(torch.equal(rows_nonfinite(out_s.to_float()), bad_rows)),
        layernorm_rows=bool(torch.equal(rows_nonfinite(ln), bad_rows)), counts=counts())
    # attention on large-magnitude q/k/v (head dim 64): softmax must stay finite, outputs bounded by max|v|
    qkv = torch.randn(576, 3 * 128, generator=g) * 200.0
    qs = ops.split_f16(qkv.to(dev))
    pad = ops.split_f16(torch.zeros(1, 3 * 128, device=dev))
    att = ops.SplitMat.empty(576, 128, dev, zero=True)
    o = ops.window_attention_split(qs, pad, 2, 24, 24, 24, 24, out=torch.empty(576, 128, device=dev), out_split=att)
    q, k, v = qkv.double().view(576, 3, 2, 64).permute(1, 2, 0, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 64 ** -0.5, -1) @ v).permute(1, 0, 2).reshape(576, 128)
    res["attention_x200"] = dict(counts=counts(), finite=bool(torch.isfinite(o).all()), rel_rmse=rel(o, ref))
    # the model end to end
    if a.model == "268":
        from cra5_amd.zoo import vaeformer_pretrained
        net, C = vaeformer_pretrained(quality=268, pretrained=False), 268
    else:
        from cra5_amd.vaeformer import VAEformer
        net, C = VAEformer(0, **synth.thin_model_kwargs()), 8
    synth.load_synthetic(net, seed=7)
    net = net.to(dev)
    x = synth.synth_frame(C, seed=2).unsqueeze(0).to(dev)
    counts()
    out = net.compress(x)
    xh = net.decompress(out["strings"], out["z_shape"])["x_hat"]
    res["model_" + a.model] = dict(counts=counts(), finite=bool(torch.isfinite(xh).all()))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
