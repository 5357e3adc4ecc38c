"""Randomised shape sweep of the GEMM / attention / LayerNorm kernels against float64 (GPU box).
   python tools/fuzz_kernels.py [seconds=60] [seed=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cra5_amd import ops
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_gemm = n_att = n_ln = 0
worst = {"gemm": 0.0, "att": 0.0, "ln": 0.0}
def rel(a, b):
    """rmse / (rms(ref) + 0.05): relative for O(1) data, absolute (the split format's 2^-25 floor, see the
    header of csrc/gemm_split_f16.hip) when the reference itself is tiny."""
    b = b.double().cpu(); a = a.double().cpu()
    return float(torch.sqrt(torch.mean((a - b) ** 2)) / (float(torch.sqrt(torch.mean(b ** 2))) + 0.05))
while time.time() < t_end:
    kind = rng.integers(0, 10)
    if kind < 6:
        M = int(rng.choice([1, 7, 33, 64, 100, 192, 255, 256, 257, 500, 648, 1000, 2048, 3000, 10368]))
        N = int(rng.choice([1, 5, 8, 31, 32, 36, 64, 77, 250, 256, 360, 512, 1024, 1080, 3072, 4096]))
        K = int(rng.choice([1, 3, 31, 32, 33, 52, 64, 100, 360, 1024, 1440, 4096, 9000]))
        if M * N * K > 6e10: continue
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * 0.05
        bias = torch.randn(N, generator=g) if rng.random() < 0.7 else None
        res = torch.randn(M, N, generator=g) if rng.random() < 0.5 else None
        gelu = bool(rng.random() < 0.4)
        ref = a.double() @ w.double().t()
        if bias is not None: ref = ref + bias.double()
        if gelu: ref = torch.nn.functional.gelu(ref)
        if res is not None: ref = ref + res.double()
        sa, sw = ops.split_f16(a.to(dev)), ops.split_f16(w.to(dev), "auto")
        mode = rng.integers(0, 3)
        out_s = ops.SplitMat.empty(M, N, dev, zero=True) if mode >= 1 else None
        if mode == 2:
            ops.gemm_nt_split(sa, sw, bias=None if bias is None else bias.to(dev), res=None if res is None else res.to(dev),
                              gelu=gelu, out_split=out_s, want_f32=False)
            out = out_s.to_float()
        else:
            pad = int(rng.choice([0, 4, 12]))
            buf = torch.full((M, N + pad), 7.0, device=dev)
            out = ops.gemm_nt_split(sa, sw, bias=None if bias is None else bias.to(dev), res=None if res is None else res.to(dev),
                                    gelu=gelu, out=buf[:, :N], out_split=out_s)
            if pad: assert bool((buf[:, N:] == 7.0).all()), ("wrote outside the view", M, N, K)
            if out_s is not None:
                e2 = rel(out_s.to_float(), ref); assert e2 < 4e-6, ("split out", M, N, K, e2)
        e = rel(out, ref); worst["gemm"] = max(worst["gemm"], e)
        assert e < 4e-6, ("gemm", M, N, K, bias is not None, res is not None, gelu, int(mode), e)
        n_gemm += 1
    elif kind < 9:
        heads = int(rng.choice([1, 2, 3])); C = 64 * heads
        wh, ww = [(4, 8), (8, 8), (24, 24), (12, 48), (48, 12), (16, 2), (32, 32)][int(rng.integers(0, 7))]
        if rng.random() < 0.3:
            H, W = wh, ww                                   # whole grid as one window
        else:
            H = int(rng.integers(1, 4)) * wh - int(rng.integers(0, wh)) % wh if rng.random() < 0.5 else wh * int(rng.integers(1, 4))
            W = ww * int(rng.integers(1, 4)) - (int(rng.integers(0, ww)) if rng.random() < 0.5 else 0)
            H, W = max(H, 1), max(W, 1)
        if not ops.split_attention_ok(C, heads, wh, ww, H, W): continue
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        qkv = torch.randn(H * W, 3 * C, generator=g) * 1.5; padrow = torch.randn(3 * C, generator=g)
        # float64 reference of the reference semantics (pad tokens carry the pad row, unmasked softmax)
        nwr, nwc = -(-H // wh), -(-W // ww)
        full = padrow.double().repeat(nwr * wh, nwc * ww, 1)
        full[:H, :W] = qkv.double().reshape(H, W, 3 * C)
        outref = torch.zeros(H, W, C, dtype=torch.float64)
        for r in range(nwr):
            for c in range(nwc):
                blk = full[r * wh:(r + 1) * wh, c * ww:(c + 1) * ww].reshape(wh * ww, 3, heads, 64)
                q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
                o = torch.softmax(q @ k.transpose(1, 2) * 0.125, -1) @ v
                o = o.transpose(0, 1).reshape(wh, ww, C)
                hh, wv = min(wh, H - r * wh), min(ww, W - c * ww)
                outref[r * wh:r * wh + hh, c * ww:c * ww + wv] = o[:hh, :wv]
        qs = ops.split_f16(qkv.to(dev)); ps = ops.split_f16(padrow.to(dev).reshape(1, -1))
        out_s = ops.SplitMat.empty(H * W, C, dev, zero=True)
        ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out_split=out_s)
        e = rel(out_s.to_float(), outref.reshape(H * W, C)); worst["att"] = max(worst["att"], e)
        assert e < 1e-5, ("attention", H, W, wh, ww, heads, e)
        n_att += 1
    else:
        rows = int(rng.choice([1, 3, 64, 648, 1000])); D = int(rng.choice([64, 360, 1024]))
        x = torch.randn(rows, D) * 3 + 1; gm = torch.randn(D); bt = torch.randn(D)
        ref = torch.nn.functional.layer_norm(x.double(), (D,), gm.double(), bt.double(), 1e-6)
        hs = ops.SplitMat.empty(rows, D, dev, zero=True)
        ops.layernorm(x.to(dev), gm.to(dev), bt.to(dev), 1e-6, out_split=hs, want_f32=False)
        e = rel(hs.to_float(), ref); worst["ln"] = max(worst["ln"], e)
        assert e < 4e-6, ("layernorm", rows, D, e)
        n_ln += 1
torch.cuda.synchronize()
print(f"fuzz ok: {n_gemm} gemm, {n_att} attention, {n_ln} layernorm cases; worst relative rmse {worst}")
