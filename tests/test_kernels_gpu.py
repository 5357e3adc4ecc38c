"""GPU parity tests of the individual HIP kernels, called through the C ABI
(cra5_amd.ops -> libcra5_amd.so), against the CPU oracle (oracle/torch_ref.py) on the
same seeded inputs.  Floating point: tolerances are written next to each check."""
import numpy as np
import pytest
import torch

from cra5_amd import ops, synth
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu


def rmse(a, b):
    return float(torch.sqrt(torch.mean((a.double().cpu() - b.double().cpu()) ** 2)))


def relerr(a, b):
    b = b.double().cpu()
    return rmse(a, b) / max(float(torch.sqrt(torch.mean(b ** 2))), 1e-30)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 64), (648, 360, 360), (1000, 1080, 360),
                                   (10368, 1024, 1024), (333, 77, 52), (2048, 4096, 1024), (70, 8192, 360)])
@pytest.mark.parametrize("epi", ["none", "bias", "gelu", "res", "gelu_res"])
def test_gemm_nt(dev, M, N, K, epi):
    if M * N * K > 2e9 and epi not in ("none", "gelu_res"):
        pytest.skip("big shape: two epilogues are enough")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t()
    if epi != "none":
        ref = ref + b.double()
    if "gelu" in epi:
        ref = torch.nn.functional.gelu(ref)
    if "res" in epi:
        ref = ref + r.double()
    out = ops.gemm_nt(a.to(dev), w.to(dev), bias=b.to(dev) if epi != "none" else None,
                      res=r.to(dev) if "res" in epi else None, gelu="gelu" in epi)
    torch.cuda.synchronize()
    # fp32 fmaf chain over K <= 1024 vs fp64: ~1e-7 relative (guide: 0.75-1.5e-7 * sum|ab|)
    assert relerr(out, ref) < 2e-6


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (10368, 1024, 1024), (2048, 4096, 1024), (1000, 360, 360),
                                   (333, 77, 52), (10368, 1024, 4096), (2048, 1024, 29480), (648, 8192, 360),
                                   # the big-tile ping-pong main loop at 1, 2, 3 and 5 k-steps (prologue / drain edges),
                                   # ragged last tile rows / columns, both tile shapes (N >= 2048: 256 x 256, else 192 x 256)
                                   (2100, 2304, 32), (2100, 2304, 64), (4099, 1030, 96), (3000, 2050, 160)])
def test_gemm_split_f16_is_fp32_accurate(dev, M, N, K):
    """The 3xf16-split MFMA GEMM against float64, side by side with the exact-f32 MFMA kernel:
    it must be at least as accurate (shorter fp32 accumulation chain), never worse than 1.5x."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * 1.7
    a[::7] *= 8.0                                  # rows with larger dynamic range
    a[1::13] *= 1e-3                               # and tiny rows (absolute-error regime of lo)
    w = torch.randn(N, K, generator=g) * 0.02
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    ad, wd, bd, rd = a.to(dev), w.to(dev), b.to(dev), r.to(dev)
    f32 = ops.gemm_nt(ad, wd, bias=bd, res=rd, gelu=True) if K % 4 == 0 else None
    sa = ops.split_f16(ad)
    sw = ops.split_f16(wd, "auto")
    assert float((sa.to_float() - ad).abs().max()) <= 2 ** -21 * float(ad.abs().max())   # 22-bit operands
    out_s = ops.SplitMat.empty(M, N, dev, zero=True)
    sp = ops.gemm_nt_split(sa, sw, bias=bd, res=rd, gelu=True, out_split=out_s)
    e_sp = rmse(sp, ref)
    e_32 = rmse(f32, ref) if f32 is not None else None
    print(f"gemm {M}x{N}x{K}: split-f16 rmse {e_sp:.2e}" + (f", exact-f32 MFMA rmse {e_32:.2e}" if e_32 else ""))
    assert relerr(sp, ref) < 2e-6
    if e_32 is not None:
        assert e_sp <= 1.5 * e_32 + 1e-9
    # the split-f16 output written by the epilogue reconstructs the fp32 output to 22 bits
    assert float((out_s.to_float() - sp).abs().max()) <= 2 ** -21 * float(sp.abs().max()) + 2 ** -24


def test_split_producers(dev):
    """LayerNorm / attention / im2col emit the same values in split-f16 as in fp32."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(648, 360, generator=g).to(dev)
    ga, be = torch.randn(360, generator=g).to(dev), torch.randn(360, generator=g).to(dev)
    y = ops.layernorm(x, ga, be)
    sm = ops.SplitMat.empty(648, 360, dev)
    ops.layernorm(x, ga, be, out_split=sm, want_f32=False)
    assert sm.Kp == 384
    assert float((sm.to_float() - y).abs().max()) <= 2 ** -21 * float(y.abs().max()) + 2 ** -24
    full = sm.data.view(torch.float16).view(648, 12, 2, 32)
    assert float(full[:, 11, :, 8:].abs().max()) == 0.0      # K padding zeroed
    qkv = torch.randn(648, 3 * 144, generator=g).to(dev)
    pad = torch.zeros(3 * 144, device=dev)
    o = ops.window_attention(qkv, pad, 2, 18, 36, 18, 36)
    so = ops.SplitMat.empty(648, 144, dev, zero=True)
    ops.window_attention(qkv, pad, 2, 18, 36, 18, 36, out_split=so, want_f32=False)
    assert float((so.to_float() - o).abs().max()) <= 2 ** -21 * float(o.abs().max()) + 2 ** -24
    img = torch.randn(3, 721, 1440, generator=g).to(dev)
    c = ops.im2col(img, 11, 10, 10, 10)
    sc = ops.SplitMat.empty(72 * 144, 330, dev, zero=True)
    ops.im2col(img, 11, 10, 10, 10, out_split=sc)
    assert float((sc.to_float() - c).abs().max()) <= 2 ** -21 * float(c.abs().max()) + 2 ** -24


@pytest.mark.parametrize("M,N,K", [(10368, 3072, 1024), (10368, 1024, 4096)])
def test_gemm_big_tiles_race_screen(dev, M, N, K):
    """The ping-pong main loop orders its LDS-DMA and fragment reads by counted waits + barriers only: an early read
    would pass whenever the DMA happens to land first.  Fifty launches under memory load (a concurrent copy stream)
    must be bit-identical."""
    g = torch.Generator().manual_seed(5)
    a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
    w = ops.split_f16((torch.randn(N, K, generator=g) * 0.03).to(dev), "auto")
    first = ops.gemm_nt_split(a, w).clone()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=dev)
    for it in range(50):
        with torch.cuda.stream(side):
            junk.mul_(1.0001)                     # HBM traffic beside the GEMM
        out = ops.gemm_nt_split(a, w)
        assert torch.equal(out, first), it
    torch.cuda.synchronize()


def test_gemm_asymmetric_layout(dev):
    """A = I with an asymmetric W catches row/col swaps of the MFMA C layout."""
    n = 128
    a = torch.eye(n)
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n)  # W[n][k]
    out = ops.gemm_nt(a.to(dev), w.to(dev))
    assert torch.equal(out.cpu(), w.t())


def test_gemm_strided_and_inplace_residual(dev):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(300, 512, generator=g).to(dev)
    a = big[:, 128:384]           # lda = 512, K = 256
    w = torch.randn(96, 256, generator=g).to(dev)
    x = torch.randn(300, 96, generator=g).to(dev)
    ref = x.double().cpu() + a.double().cpu() @ w.double().cpu().t()
    ops.gemm_nt(a, w, res=x, out=x)   # in-place residual update
    assert relerr(x, ref) < 2e-6


@pytest.mark.parametrize("rows,D", [(10368, 1024), (648, 360), (100, 128), (7, 144), (33, 2048)])
def test_layernorm(dev, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 1.5
    ga = torch.randn(D, generator=g)
    be = torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), ga.double(), be.double(), 1e-6)
    out = ops.layernorm(x.to(dev), ga.to(dev), be.to(dev), 1e-6)
    assert rmse(out, ref) < 1e-6  # fp32 rounding of O(1) values


def _attn_gpu(x, sd, pre, heads, H, W, ws, dev):
    qkv = ops.gemm_nt(x.to(dev), sd[pre + ".qkv.weight"].to(dev), bias=sd[pre + ".qkv.bias"].to(dev))
    wh, ww = ws if ws is not None else (H, W)
    o = ops.window_attention(qkv, sd[pre + ".qkv.bias"].to(dev), heads, H, W, wh, ww)
    return ops.gemm_nt(o, sd[pre + ".proj.weight"].to(dev), bias=sd[pre + ".proj.bias"].to(dev))


@pytest.mark.parametrize("ws", [(24, 24), (12, 48), (48, 12), None])
def test_window_attention_hd64(dev, ws):
    H, W, C, heads = 72, 144, 128, 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, H * W, C, generator=g)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C),
              "attn.proj.bias": (C,)}
    sd = synth.fill_state_dict(shapes, seed=21)
    sd["attn.qkv.weight"] *= 3.0  # sharper softmax
    xd = {k: v.double() for k, v in sd.items()}
    if ws is None:
        ref = R.attention_global(x.double(), xd, "attn", heads)
    else:
        ref = R.attention_window(x.double(), xd, "attn", heads, H, W, ws)
    out = _attn_gpu(x[0], sd, "attn", heads, H, W, ws, dev)
    # tolerance: the reference's own fp32 path is 2.3e-6 RMSE from fp64 on this input
    # (outputs O(1.2)); allow 4e-6
    assert rmse(out, ref[0]) < 4e-6, rmse(out, ref[0])


@pytest.mark.parametrize("ws", [(24, 24), (12, 48), (48, 12), None])
def test_window_attention_split_f16(dev, ws):
    """f16-split MFMA attention (split qkv in, split out) vs float64; must match the exact-f32
    kernel's accuracy class."""
    H, W, C, heads = 72, 144, 128, 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, H * W, C, generator=g)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C),
              "attn.proj.bias": (C,)}
    sd = synth.fill_state_dict(shapes, seed=21)
    sd["attn.qkv.weight"] *= 3.0
    xd = {k: v.double() for k, v in sd.items()}
    ref = R.attention_global(x.double(), xd, "attn", heads) if ws is None else \
        R.attention_window(x.double(), xd, "attn", heads, H, W, ws)
    wh, ww = ws if ws is not None else (H, W)
    xs = ops.split_f16(x[0].to(dev))
    qkv_s = ops.SplitMat.empty(H * W, 3 * C, dev)
    ops.gemm_nt_split(xs, ops.split_f16(sd["attn.qkv.weight"].to(dev), "auto"), bias=sd["attn.qkv.bias"].to(dev),
                      out_split=qkv_s, want_f32=False)
    pad_s = ops.split_f16(sd["attn.qkv.bias"].to(dev).reshape(1, -1))
    att = ops.SplitMat.empty(H * W, C, dev, zero=True)
    ops.window_attention_split(qkv_s, pad_s, heads, H, W, wh, ww, out_split=att)
    out = ops.gemm_nt_split(att, ops.split_f16(sd["attn.proj.weight"].to(dev), "auto"), bias=sd["attn.proj.bias"].to(dev))
    e_split = rmse(out, ref[0])
    e_f32 = rmse(_attn_gpu(x[0], sd, "attn", heads, H, W, ws, dev), ref[0])
    print(f"attention {ws}: split-f16 rmse {e_split:.2e}, exact-f32 rmse {e_f32:.2e}")
    assert e_split < 4e-6 and e_split <= 1.5 * e_f32 + 1e-8


def _bal_plan(T, groups, nw=12):
    """Host restatement of balanced_plan (csrc/attention_split_f16.hip): (full wave-tiles, groups of the key-split part,
    most pieces overlapping a group)."""
    n_full = T // (groups * nw)
    rem = T - n_full * groups * nw
    n_grp = (rem + nw - 1) // nw
    S = n_grp * T

    def piece_of(s):
        c = min(groups - 1, s * groups // S)
        while c > 0 and S * c // groups > s:
            c -= 1
        while c < groups - 1 and S * (c + 1) // groups <= s:
            c += 1
        return c
    maxp = max((piece_of((g + 1) * T - 1) - piece_of(g * T) + 1 for g in range(n_grp)), default=0)
    return n_full * groups * nw, n_grp, maxp


def test_global_attention_balanced_schedule(dev):
    """The balanced whole-grid schedule (12-wave work-groups throughout: one pass of full wave-tiles per slot, the remaining
    tiles' key loops laid end to end and cut into one piece per slot, partial softmaxes merged by attention_merge_kernel)
    at the model's shape, 10 368 tokens x 16 heads: queries of the full pass run the SAME key loop as the plain launch ->
    bit-identical; the others are merged from 2-3 key ranges -> compared with float64, same accuracy class as the plain
    kernel."""
    H, W, C, heads = 72, 144, 1024, 16
    N = H * W
    ok, nb = ops.attention_balanced_plan(N, heads)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if not ok:
        pytest.skip(f"no balanced plan on a {cus}-CU device")
    full_tiles, n_grp, maxp = _bal_plan(N // 32, cus // heads)
    assert nb == heads * n_grp * maxp * 12 * 32 * 68 * 4 == ops.attention_workspace_bytes(N, heads)
    assert cus != 256 or (full_tiles, n_grp, maxp) == (192, 11, 3)      # 256 CUs: 16 slots x 12 full tiles + 11 groups
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(N, 3 * C, generator=g)
    qkv[:, :2 * C] *= 1.7                                   # logits of a few units: a softmax that is not flat
    qkv[777, C:2 * C] *= 3.0                                # ... and one dominant key
    qs = ops.split_f16(qkv.to(dev))
    pad = ops.split_f16(torch.zeros(1, 3 * C, device=dev))
    plain = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ws.fill_(0xFF)                                                             # NaN patterns: every partial read was written
    out = torch.full((N, C), float("nan"), device=dev)
    out_s = ops.SplitMat.empty(N, C, dev, zero=True)
    ops.window_attention_split(qs, pad, heads, H, W, H, W, out=out, out_split=out_s, workspace=ws)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()                                          # every token written exactly once
    n_full = full_tiles * 32
    assert torch.equal(out[:n_full], plain[:n_full])
    # the key-split part against float64 (a 1/8 sample of its tokens: the fp64 softmax over 10 368 keys is host work)
    tail = torch.arange(n_full, N, 8)
    qd = qkv.double().view(N, 3, heads, 64)
    q, k, v = qd[tail, 0].permute(1, 0, 2), qd[:, 1].permute(1, 0, 2), qd[:, 2].permute(1, 0, 2)
    ref = (torch.softmax((q * 64 ** -0.5) @ k.transpose(-1, -2), -1) @ v).permute(1, 0, 2).reshape(tail.numel(), C)
    e_bal, e_plain = rmse(out[tail.to(dev)], ref), rmse(plain[tail.to(dev)], ref)
    print(f"balanced global attention: key-split tokens rmse {e_bal:.2e} (plain kernel {e_plain:.2e})")
    assert e_bal < 2e-6 and e_bal <= 1.5 * e_plain + 1e-8
    assert rmse(out[n_full:], plain[n_full:]) < 1e-6
    assert rmse(out_s.to_float(), out) < 1e-6                                  # split output = fp32 output (22 bits)
    # deterministic: the merge adds the key ranges in a fixed order
    out2 = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev), workspace=ws)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("H,W,heads", [(60, 120, 16), (64, 64, 32), (48, 160, 16), (48, 128, 16), (96, 128, 32)])
def test_global_attention_balanced_schedule_other_shapes(dev, H, W, heads):
    """Other plans of the balanced schedule on 256 CUs: 7200 tokens x 16 heads (225 wave-tiles = 192 full + 33 in groups
    of 12, 12, 9: a partial last group), 4096 x 32 (8 slots: 96 full + 32), 7680 x 16 (192 + 48), 6144 x 16 (192 full
    tiles exactly: NO key-split part and no workspace), 12 288 x 32 (384 tiles = 4 full passes of 8 slots) - against the
    plain launch: identical on the full-pass tokens, within fp32 noise on the key-split ones."""
    C, N = 64 * heads, H * W
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if cus != 256:
        pytest.skip("plans below are written out for 256 CUs")
    ok, nb = ops.attention_balanced_plan(N, heads)
    full_tiles, n_grp, maxp = _bal_plan(N // 32, cus // heads)
    assert ok and nb == heads * n_grp * maxp * 12 * 32 * 68 * 4
    g = torch.Generator().manual_seed(H * 1000 + W)
    qkv = torch.randn(N, 3 * C, generator=g)
    qkv[:, :2 * C] *= 1.5
    qs = ops.split_f16(qkv.to(dev))
    pad = ops.split_f16(torch.zeros(1, 3 * C, device=dev))
    plain = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev).fill_(0xFF) if nb else None
    out = torch.full((N, C), float("nan"), device=dev)
    ops.window_attention_split(qs, pad, heads, H, W, H, W, out=out, workspace=ws, balanced=True)   # (nb == 0: no workspace)
    torch.cuda.synchronize()
    n_full = full_tiles * 32
    assert torch.isfinite(out).all() and torch.equal(out[:n_full], plain[:n_full])
    assert (nb == 0) == (n_full == N)
    if n_full < N:
        e = rmse(out[n_full:], plain[n_full:])
        print(f"balanced attention {H}x{W}, {heads} heads: {N // 32 - full_tiles} key-split tile(s) in {n_grp} group(s), "
              f"rmse vs plain {e:.2e}")
        assert e < 1e-6


@pytest.mark.parametrize("boost", [4.0, 0.5, 1.0])
def test_attention_split_softmax_spike(dev, boost):
    """Late dominant key.  boost 4: its score is ~46 log2 units above the running max -> the O / l rescale branch is
    taken; boost 0.5 / 1: ~6 / ~11.5 units -> below / just above the deferral threshold (2^8): below it the running
    max stays put and that key's p = exp2(s - m) is ~55 (> 1), exact in the hi / lo split."""
    H, W, C, heads = 8, 72, 64, 1
    N = H * W
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(N, 3 * C, generator=g)
    qkv[:, C:2 * C] *= 0.1
    qkv[500, C:2 * C] = qkv[17, :C] * boost
    q, k, v = qkv[:, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
    ref = torch.softmax((q * C ** -0.5) @ k.t(), -1) @ v
    qs = ops.split_f16(qkv.to(dev))
    pad = ops.split_f16(torch.zeros(1, 3 * C, device=dev))
    out = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev))
    assert rmse(out, ref) < 2e-6


@pytest.mark.parametrize("case", ["boost4", "boost0.5", "boost1", "low_start", "ramp"])
def test_attention_hi_only_reference_point(dev, case):
    """Reduced-precision attention (hi_only): the exponent's reference point lives in the score MFMA's C operand
    (p = exp2(acc) with Q pre-scaled by scale * log2 e) and is moved by an MFMA when a tile's row sum shows that some p left 2^13 (P_SAFE; the fp32-accurate form defers at 2^8).
    Late dominant keys (as the fp32-accurate twin above), a first tile far BELOW zero (the first tile sets the
    reference in either direction) and a steady ramp (many small moves) against float64 on the f16-rounded operands."""
    H, W, C, heads = 8, 72, 64, 1
    N = H * W
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(N, 3 * C, generator=g)
    qkv[:, C:2 * C] *= 0.1
    if case.startswith("boost"):
        qkv[500, C:2 * C] = qkv[17, :C] * float(case[5:])
    elif case == "low_start":
        qkv[:, :C] = qkv[:, :C].abs()                # every score of the first key tiles ~ -60 (log2 domain: -87)
        qkv[:64, C:2 * C] = -1.0
    else:
        qkv[:, :C] = qkv[:, :C].abs()
        qkv[:, C:2 * C] = (torch.arange(N).float() / N * 3.0).reshape(N, 1)   # scores grow by ~0.2 log2 units per tile... x 64 d
    x16 = qkv.half().double()
    q, k, v = x16[:, :C], x16[:, C:2 * C], x16[:, 2 * C:]
    ref = torch.softmax((q * C ** -0.5) @ k.t(), -1) @ v
    qs = ops.split_f16(qkv.to(dev))
    pad = ops.split_f16(torch.zeros(1, 3 * C, device=dev))
    out = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev), hi_only=True)
    e, r = rmse(out, ref), float(torch.sqrt(torch.mean(ref ** 2)))
    print(f"hi_only attention, {case}: rmse {e:.2e} (rms of the output {r:.2e})")
    assert torch.isfinite(out).all() and e < 1.5e-3 * max(r, 0.05)


def test_global_attention_hd72_ragged(dev):
    """648 tokens (not a multiple of the 32-key tile) and head dim 72: hyper-prior shape."""
    H, W, C, heads = 18, 36, 144, 2
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, H * W, C, generator=g)
    shapes = {"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C),
              "attn.proj.bias": (C,)}
    sd = synth.fill_state_dict(shapes, seed=22)
    sd["attn.qkv.weight"] *= 3.0
    ref = R.attention_global(x.double(), {k: v.double() for k, v in sd.items()}, "attn", heads)
    out = _attn_gpu(x[0], sd, "attn", heads, H, W, None, dev)
    assert rmse(out, ref[0]) < 4e-6


def test_attention_softmax_spike(dev):
    """One key dominates late in the sequence: forces the online-softmax rescale path."""
    H, W, C, heads = 8, 72, 64, 1
    N = H * W
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(N, 3 * C, generator=g)
    qkv[:, C:2 * C] *= 0.1
    qkv[500, C:2 * C] = qkv[17, :C] * 4.0   # key 500 matches query 17 strongly
    q, k, v = qkv[:, :C].double(), qkv[:, C:2 * C].double(), qkv[:, 2 * C:].double()
    ref = torch.softmax((q * C ** -0.5) @ k.t(), -1) @ v
    out = ops.window_attention(qkv.to(dev), torch.zeros(3 * C, device=dev), heads, H, W, H, W)
    assert rmse(out, ref) < 2e-6


def test_im2col_col2im(dev):
    g = torch.Generator().manual_seed(4)
    C, H, W = 5, 721, 1440
    x = torch.randn(C, H, W, generator=g)
    mean = torch.randn(C, generator=g)
    std = torch.rand(C, generator=g) + 0.5
    cols = ops.im2col(x.to(dev), 11, 10, 10, 10, ldk=576, mean=mean.to(dev), std=std.to(dev))
    xn = (x - mean[:, None, None]) / std[:, None, None]
    ref = torch.nn.functional.unfold(xn[None], (11, 10), stride=(10, 10))[0].t()  # [tokens, C*110]
    assert torch.equal(cols[:, :550].cpu(), ref)            # bit-exact: same IEEE ops
    assert torch.count_nonzero(cols[:, 550:]) == 0
    # overlap-add back (ConvTranspose2d with identity "weights")
    back = ops.col2im(cols[:, :550], C, 11, 10, 10, 10, 72, 144)
    ref_back = torch.nn.functional.fold(ref.t()[None], (H, W), (11, 10), stride=(10, 10))[0]
    assert rmse(back, ref_back) < 1e-7
    den = ops.col2im(cols[:, :550], C, 11, 10, 10, 10, 72, 144, mean=mean.to(dev), std=std.to(dev))
    assert rmse(den, ref_back * std[:, None, None] + mean[:, None, None]) < 1e-6


def test_transpose_pixel_shuffle(dev):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(10368, 256, generator=g)
    assert torch.equal(ops.transpose(x.to(dev)).cpu(), x.t())
    x = torch.randn(77, 45, generator=g)
    assert torch.equal(ops.transpose(x.to(dev)).cpu(), x.t())
    lin = torch.randn(18 * 36, 4 * 4 * 32, generator=g)
    from einops import rearrange
    ref = rearrange(lin.reshape(1, 18, 36, -1), "b h w (p1 p2 c) -> b c (h p1) (w p2)", p1=4, p2=4)[0]
    assert torch.equal(ops.pixel_shuffle(lin.to(dev), 18, 36, 4, 4).cpu(), ref)


def test_gaussian_conditional(dev):
    g = torch.Generator().manual_seed(8)
    n = (1, 16, 72, 144)
    y = torch.randn(n, generator=g) * 3
    means = torch.randn(n, generator=g)
    scales = torch.randn(n, generator=g).abs() * 4 - 0.5
    table = R.get_scale_table()
    scales.view(-1)[:64] = table            # exact table hits (<= comparison edge)
    scales.view(-1)[64:128] = table * (1 + 1e-7)
    o = ops.gaussian_conditional(scales.to(dev), means.to(dev), table.to(dev), y=y.to(dev),
                                 want=("idx", "sym", "y_hat", "lik"))
    assert torch.equal(o["idx"].cpu(), R.gc_build_indexes(scales, table))       # integers: bit-exact
    assert torch.equal(o["sym"].cpu(), R.gc_symbols(y, means))
    y_hat, lik = R.gc_forward(y, scales, means)
    assert torch.equal(o["y_hat"].cpu(), y_hat)
    assert float((o["lik"].cpu() - lik).abs().max()) < 2e-7                       # erfc implementation noise
    d = ops.gaussian_conditional(scales.to(dev), means.to(dev), table.to(dev), sym_in=o["sym"], want=("y_hat",))
    assert torch.equal(d["y_hat"].cpu(), y_hat)


def test_entropy_bottleneck(dev):
    from cra5_amd.entropy import eb_pack_params
    C = 16
    shapes = {f"entropy_bottleneck._matrix{i}": s for i, s in enumerate([(C, 3, 1), (C, 3, 3), (C, 3, 3), (C, 3, 3), (C, 1, 3)])}
    shapes.update({f"entropy_bottleneck._bias{i}": s for i, s in enumerate([(C, 3, 1)] * 4 + [(C, 1, 1)])})
    shapes.update({f"entropy_bottleneck._factor{i}": (C, 3, 1) for i in range(4)})
    shapes["entropy_bottleneck.quantiles"] = (C, 1, 3)
    sd = synth.fill_state_dict(shapes, seed=9)
    g = torch.Generator().manual_seed(10)
    z = torch.randn(1, C, 18, 36, generator=g) * 5
    z_hat, lik = R.eb_forward(z, sd)
    sym = R.eb_symbols(z, sd)
    med = R.eb_medians(sd)
    o = ops.entropy_bottleneck(med.to(dev), eb_pack_params(sd).to(dev), z=z[0].reshape(C, -1).contiguous().to(dev),
                               want=("sym", "z_hat", "lik"))
    assert torch.equal(o["sym"].cpu().reshape(sym.shape[1:]), sym[0])
    assert torch.equal(o["z_hat"].cpu().reshape(z_hat.shape[1:]), z_hat[0])
    assert float((o["lik"].cpu().reshape(lik.shape[1:]) - lik[0]).abs().max()) < 5e-7


def test_gdn_golden(dev, golden_dir):
    d = np.load(f"{golden_dir}/ops_small.npz")
    for inv in (0, 1):
        x = torch.from_numpy(d[f"gdn_inv{inv}_x"])
        beta_p, gamma_p = torch.from_numpy(d[f"gdn_inv{inv}_beta"]), torch.from_numpy(d[f"gdn_inv{inv}_gamma"])
        ped = (2 ** -18) ** 2
        beta = torch.clamp(beta_p, min=(1e-6 + ped) ** 0.5) ** 2 - ped
        gamma = torch.clamp(gamma_p, min=ped ** 0.5) ** 2 - ped
        y = ops.gdn(x.to(dev), beta.to(dev), gamma.contiguous().to(dev), inverse=bool(inv))
        assert rmse(y, torch.from_numpy(d[f"gdn_inv{inv}_y"])) < 1e-6


def test_gdn_module_state_dict_and_forward(dev, golden_dir):
    from cra5_amd.layers import GDN
    d = np.load(f"{golden_dir}/ops_small.npz")
    for inv in (0, 1):
        m = GDN(12, inverse=bool(inv))
        assert set(m.state_dict()) == {"beta", "gamma", "beta_reparam.pedestal", "beta_reparam.lower_bound.bound",
                                       "gamma_reparam.pedestal", "gamma_reparam.lower_bound.bound"}
        m.beta.data = torch.from_numpy(d[f"gdn_inv{inv}_beta"])
        m.gamma.data = torch.from_numpy(d[f"gdn_inv{inv}_gamma"])
        y = m.to(dev)(torch.from_numpy(d[f"gdn_inv{inv}_x"]).to(dev))
        assert rmse(y, torch.from_numpy(d[f"gdn_inv{inv}_y"])) < 1e-6


@pytest.mark.parametrize("C", [8, 7, 13])
def test_tiled_patch_gather_scatter_matches_generic(dev, C):
    """The LDS-tiled ERA5-geometry kernels (taken for k=(11,10), s=(10,10)) against torch
    unfold / fold, incl. a ragged last channel chunk, fused (de)normalisation and split output."""
    g = torch.Generator().manual_seed(40 + C)
    H, W = 721, 1440
    x = torch.randn(C, H, W, generator=g)
    mean = torch.randn(C, generator=g)
    std = torch.rand(C, generator=g) + 0.5
    K = C * 110
    Kp = (K + 31) // 32 * 32
    xn = (x - mean[:, None, None]) / std[:, None, None]
    ref = torch.nn.functional.unfold(xn[None], (11, 10), stride=(10, 10))[0].t().contiguous()
    cols = ops.im2col(x.to(dev), 11, 10, 10, 10, ldk=Kp, mean=mean.to(dev), std=std.to(dev))
    assert torch.equal(cols[:, :K].cpu(), ref) and torch.count_nonzero(cols[:, K:]) == 0
    sm = ops.SplitMat.empty(72 * 144, K, dev, zero=True)
    ops.im2col(x.to(dev), 11, 10, 10, 10, mean=mean.to(dev), std=std.to(dev), out_split=sm)
    assert float((sm.to_float().cpu() - ref).abs().max()) <= 2 ** -21 * float(ref.abs().max()) + 2 ** -24
    # reduced-precision mode: PLAIN rows = the hi plane, the K padding re-zeroed by the kernel (the buffer holds junk here)
    sp = ops.SplitMat.empty(72 * 144, K, dev)
    sp.data.fill_(0x3c00)
    ops.im2col(x.to(dev), 11, 10, 10, 10, mean=mean.to(dev), std=std.to(dev), out_split=sp, out_plain=True)
    assert sp.plain and torch.equal(sp.data[:, :K], _hi_plane(sm)[:, :K]) and torch.count_nonzero(sp.data[:, K:sp.Kp]) == 0
    assert bool((sp.data[:, sp.Kp:] == 0x3c00).all())          # nothing written past the plain row
    back = ops.col2im(cols[:, :K], C, 11, 10, 10, 10, 72, 144, mean=mean.to(dev), std=std.to(dev))
    ref_back = torch.nn.functional.fold(ref.t()[None], (H, W), (11, 10), stride=(10, 10))[0]
    assert rmse(back, ref_back * std[:, None, None] + mean[:, None, None]) < 1e-6
    raw = ops.col2im(cols[:, :K], C, 11, 10, 10, 10, 72, 144)
    assert torch.equal(raw.cpu(), ref_back)   # two-term sums: bit-exact


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (2048, 4096, 1024), (10368, 1024, 1024), (648, 360, 360),
                                   (2048, 2048, 96), (1100, 2304, 160), (2048, 1024, 8192 + 64 * 5), (1536, 1024, 29480)])
def test_gemm_hi_only_is_plain_f16(dev, M, N, K):
    """CRA5_GEMM_HI_ONLY (reduced-precision mode, BASELINE.json configs[4]): the product of the f16-rounded
    operands with fp32 accumulation - compared against exactly that in float64, and an order of magnitude
    away from the fp32-accurate result so that the flag is known to take effect.  Big tiles with Kp % 64 == 0 take the
    wide form (64 k-values per k-step: the hi halves of two chunks per LDS row); Kp = 96 / 160 the 32-wide one; K >
    8192 is chained in chunks that are multiples of 64."""
    g = torch.Generator().manual_seed(7 * M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.02
    b = torch.randn(N, generator=g)
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    sa, sw = ops.split_f16(ad), ops.split_f16(wd, "auto")
    out = ops.gemm_nt_split(sa, sw, bias=bd, hi_only=True)
    a16 = a.half().double()
    w16 = ((w * (1.0 / sw.scale_inv)).half().double()) * sw.scale_inv
    ref16 = a16 @ w16.t() + b.double()
    ref32 = a.double() @ w.double().t() + b.double()
    assert relerr(out, ref16) < 3e-6
    e = relerr(out, ref32)
    assert 1e-5 < e < 5e-3, e


def _hi_plane(sm):
    """the hi halves of a split matrix as a [rows, Kp] int16 tensor"""
    return sm.data.view(sm.rows, sm.Kp // 32, 2, 32)[:, :, 0].reshape(sm.rows, sm.Kp)


def _plain_rows_of(sm_split_layout_buffer):
    """a plain matrix living in the first half of every row of a split-layout buffer (what the model's workspaces hold)"""
    sm = ops.SplitMat.empty(sm_split_layout_buffer.rows, sm_split_layout_buffer.K, sm_split_layout_buffer.data.device, zero=True)
    sm.data[:, : sm.Kp] = _hi_plane(sm_split_layout_buffer)
    sm.plain = True
    return sm


@pytest.mark.parametrize("M,N,K,epi", [(2048, 4096, 1024, "gelu_out"), (10368, 1024, 1024, "res"), (10368, 3072, 1024, "bias_out"),
                                       (2048, 2048, 8192 + 64 * 5, "res"), (4096, 1024, 29480, "bias")])
def test_gemm_plain_f16_operands_equal_the_split_hi_planes(dev, M, N, K, epi):
    """Round 5, reduced-precision mode: CRA5_GEMM_A_PLAIN / _W_PLAIN / _OUT_PLAIN (a row = contiguous halves: one full
    128-byte line per 64-wide k-step) against the same launch on split rows' hi planes - the SAME arithmetic in the same
    order: fp32 outputs bit-identical, the plain output row == the hi plane of the split output.  Every combination of
    plain / split A and W; long K chained in 64-multiples."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if epi == "res" else None
    sa, sw = ops.split_f16(a), ops.split_f16(w, "auto")
    pa, pw = _plain_rows_of(sa), sw.plain_copy()
    assert torch.equal(pw.data, _hi_plane(sw)) and pw.plain and pw.pitch == sw.Kp and pa.pitch == 2 * sa.Kp
    kw = dict(bias=b, res=r, gelu="gelu" in epi, hi_only=True)
    if epi.endswith("_out"):
        ref = ops.SplitMat.empty(M, N, dev, zero=True)
        ops.gemm_nt_split(sa, sw, out_split=ref, want_f32=False, **kw)
        for A_, W_ in ((pa, pw), (sa, pw), (pa, sw)):
            got = ops.SplitMat.empty(M, N, dev, zero=True)
            ops.gemm_nt_split(A_, W_, out_split=got, want_f32=False, out_plain=True, **kw)
            assert got.plain and torch.equal(got.data[:, : got.Kp], _hi_plane(ref))
            assert torch.count_nonzero(got.data[:, got.Kp:]) == 0          # nothing written past the plain row
            assert torch.equal(got.to_float(), ref.data.view(torch.float16).view(M, -1, 2, 32)[:, :, 0].reshape(M, -1)[:, :N].float())
    else:
        ref = ops.gemm_nt_split(sa, sw, **kw)
        for A_, W_ in ((pa, pw), (sa, pw), (pa, sw)):
            assert torch.equal(ops.gemm_nt_split(A_, W_, **kw), ref)
    # the flags are refused where the wide form does not run (fp32-accurate mode, small launches)
    with pytest.raises(Exception):
        ops.gemm_nt_split(pa, pw, bias=b)
    with pytest.raises(Exception):
        ops.gemm_nt_split(_plain_rows_of(ops.split_f16(a[:256])), pw, hi_only=True)


def test_layernorm_plain_rows(dev):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(1000, 1024, generator=g) * 3 + 0.5).to(dev)
    gam, bet = (1 + 0.1 * torch.randn(1024, generator=g)).to(dev), (0.1 * torch.randn(1024, generator=g)).to(dev)
    s_split, s_plain = ops.SplitMat.empty(1000, 1024, dev, zero=True), ops.SplitMat.empty(1000, 1024, dev, zero=True)
    ops.layernorm(x, gam, bet, 1e-6, out_split=s_split, want_f32=False)
    ops.layernorm(x, gam, bet, 1e-6, out_split=s_plain, want_f32=False, out_plain=True)
    assert s_plain.plain and not s_split.plain
    assert torch.equal(s_plain.data[:, :1024], _hi_plane(s_split)) and torch.count_nonzero(s_plain.data[:, 1024:]) == 0


@pytest.mark.parametrize("ws", [(24, 24), (48, 12), None])
def test_window_attention_plain_rows_equal_the_hi_only_split_launch(dev, ws):
    """hi_only = 3 in the C ABI: plain qkv / pad rows in, plain rows out - bit-identical to the reduced-precision launch on
    split rows (same f16 values, same order), for windows (incl. the padded shape), the plain whole-grid launch and the
    balanced one with its merge kernel."""
    H, W, C, heads = 72, 144, 128, 2
    g = torch.Generator().manual_seed(12)
    qkv = torch.randn(H * W, 3 * C, generator=g).to(dev)
    bias = torch.randn(1, 3 * C, generator=g).to(dev)
    qs, ps = ops.split_f16(qkv), ops.split_f16(bias)
    qp, pp = _plain_rows_of(qs), ps.plain_copy()
    wh, ww = ws if ws is not None else (H, W)
    for balanced in ((False, True) if ws is None else (False,)):
        wsb = None
        if balanced:
            ok, nb = ops.attention_balanced_plan(H * W, heads)
            if not ok:
                continue
            wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        o_s, o_p = ops.SplitMat.empty(H * W, C, dev, zero=True), ops.SplitMat.empty(H * W, C, dev, zero=True)
        f_s = torch.empty(H * W, C, device=dev)
        f_p = torch.empty(H * W, C, device=dev)
        ops.window_attention_split(qs, ps, heads, H, W, wh, ww, out=f_s, out_split=o_s, hi_only=True, workspace=wsb, balanced=balanced or None)
        ops.window_attention_split(qp, pp, heads, H, W, wh, ww, out=f_p, out_split=o_p, hi_only=True, workspace=wsb, balanced=balanced or None)
        assert o_p.plain and not o_s.plain
        assert torch.equal(f_p, f_s)
        assert torch.equal(o_p.data[:, :C], _hi_plane(o_s)) and torch.count_nonzero(o_p.data[:, C:]) == 0


def test_split_attention_shape_rule(dev):
    """Windows that are not the whole grid keep a per-block row-offset table: more than
    ops.MAX_WIN_TOKENS tokens per window is refused by the C ABI (the model then uses the exact-f32
    kernel, ops.split_attention_ok says so beforehand)."""
    H, W, C, heads = 64, 64, 64, 1
    assert ops.split_attention_ok(C, heads, 32, 32, H, W)           # 1024-token windows: fine
    assert not ops.split_attention_ok(C, heads, 64, 32, H, W)       # 2048-token windows: no
    assert ops.split_attention_ok(C, heads, 64, 64, H, W)           # the whole grid: fine
    qkv = ops.split_f16(torch.randn(H * W, 3 * C, device=dev))
    pad = ops.split_f16(torch.randn(1, 3 * C, device=dev))
    out = ops.SplitMat.empty(H * W, C, dev, zero=True)
    with pytest.raises(Exception):
        ops.window_attention_split(qkv, pad, heads, H, W, 64, 32, out_split=out)
    ops.window_attention_split(qkv, pad, heads, H, W, 64, 64, out_split=out)
    ops.window_attention_split(qkv, pad, heads, H, W, 32, 32, out_split=out)
    torch.cuda.synchronize()
    assert torch.isfinite(out.to_float()).all()


def test_split_f16_range_guard(dev):
    """Range safety of the split-f16 engine (VERDICT r1 / ADVICE r1, reworked in round 4): an operand beyond f16's
    65 504 is never clipped quietly - the split poisons it (hi = +-inf, lo = -+inf -> NaN products), so exactly the
    GEMM rows that read it come out non-finite (the model's range guard then re-runs the frame on the exact-f32 engines:
    tests/test_model_gpu.py::test_range_guard_*); the `rangecheck` build flavour counts them.  Runs tools/range_audit.py against that flavour in a subprocess:
    activations scaled by 1e-6 / 1 / 1e4 / 1e5 through split -> GEMM (+ split output) and LayerNorm,
    attention on |q.k| ~ 1e5 logits, and the thin model end to end (must report zero events)."""
    import json
    import os
    import subprocess
    import sys
    from cra5_amd import build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libp = B.build(flavour="rangecheck")    # no-op when the in-tree flavour is newer than its sources
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "range_audit.py"), "--model", "thin"],
                       env=dict(os.environ, CRA5_LIB=libp), cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(res, indent=1))
    for name in ("1e-6", "1", "1e4"):
        c = res[name]
        assert c["split_in"] == [0, 0] and c["gemm_out"] == [0, 0] and c["layernorm"] == [0, 0], (name, c)
        assert c["gemm_finite"] and c["ln_finite"]
        if name != "1e-6":   # (var + eps)^-1/2 with eps = 1e-6 is not scale-free at |x| ~ 1e-6: rms 1e-3 is right
            assert abs(c["ln_rms"] - 1.0) < 1e-3
    assert res["1"]["gemm_rel_rmse"] < 2e-6 and res["1e4"]["gemm_rel_rmse"] < 2e-6
    # 1e-6-scaled activations sit in the ABSOLUTE-error regime of the lo plane (f16 subnormals, 2^-25):
    # the products are ~1e-7 and carry an absolute error of ~1e-9 -> percent-level relative error.
    # Stated, not hidden: fp32-class accuracy needs |x| >~ 2^-3 * 2^-11 (gemm_split_f16.hip:12-14).
    assert res["1e-6"]["gemm_rel_rmse"] < 0.2
    # 1e5: ~50 % of N(0, 1e5) exceeds 65504 -> counted, and every GEMM row that holds one is poisoned - none clipped
    c = res["1e5"]
    assert c["split_in"][0] > 100000 and c["split_in"][1] == 0
    assert c["rows_over"] > 0 and c["poisoned_equals_over"] and not c["gemm_finite"], c
    assert c["ln_finite"]        # LayerNorm normalises in fp32 before its split store: scale-free
    assert res["attention_x200"]["finite"] and res["attention_x200"]["counts"] == [0, 0]
    # logits of +-3e5: the softmax is one-hot and an ulp of a logit moves whole rows -> 1e-4-class, finite
    assert res["attention_x200"]["rel_rmse"] < 1e-3
    assert res["model_thin"]["counts"] == [0, 0] and res["model_thin"]["finite"]
    # NaN / inf are NOT saturated: they reach the outputs (v_med3 alone turned NaN into -65504)
    nf = res["nonfinite"]
    assert nf["split_keeps"] and nf["gemm_rows"] and nf["gemm_split_rows"] and nf["layernorm_rows"], nf
    assert nf["counts"][1] >= 2     # and the rangecheck flavour counts them as non-finite events


@pytest.mark.parametrize("M,N,K", [(648, 1080, 360), (648, 360, 360), (648, 1440, 360), (648, 360, 1440),
                                   (648, 360, 4096), (648, 256, 360), (648, 256, 256), (162, 432, 144),
                                   (648, 8192, 360), (100, 77, 52), (2000, 3000, 96)])
def test_small_gemm_split_matches_fp64_and_big_engine(dev, M, N, K):
    """csrc/hyper.hip small-M GEMM (all dispatch branches: 1 / 4 / 8-way in-block split-K, 32 / 64 / 128-column
    wave tiles) against float64 and against the big-tile engine, with every epilogue flag and the
    split-f16 output incl. its zero pad columns."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = torch.randn(M, K, generator=g) * 1.3
    w = torch.randn(N, K, generator=g) * (1.0 / np.sqrt(K))
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    sa, sw = ops.split_f16(a.to(dev)), ops.split_f16(w.to(dev), "auto")
    out_s = ops.SplitMat.empty(M, N, dev)          # NOT zero-initialised: the kernel must write the pad
    out_s.data.fill_(0x7e00)                        # f16 NaN pattern everywhere
    out = ops.small_gemm_nt_split(sa, sw, bias=b.to(dev), res=r.to(dev), gelu=True, out_split=out_s)
    big = ops.gemm_nt_split(sa, sw, bias=b.to(dev), res=r.to(dev), gelu=True)
    e, e_big = relerr(out, ref), relerr(big, ref)
    print(f"small gemm {M}x{N}x{K}: rel {e:.2e} (big engine {e_big:.2e})")
    assert e < 2e-6 and e <= 1.5 * e_big + 1e-8
    assert float((out_s.to_float() - out).abs().max()) <= 2 ** -21 * float(out.abs().max()) + 2 ** -24
    raw = out_s.data.view(torch.float16).view(M, out_s.Kp // 32, 2, 32)
    if out_s.Kp > N:
        pad = raw.permute(0, 1, 3, 2).reshape(M, out_s.Kp, 2)[:, N:, :]
        assert float(pad.float().abs().max()) == 0.0      # pad columns: zeros, not NaN
    # plain (no epilogue) + determinism: two launches are bit-identical
    p1 = ops.small_gemm_nt_split(sa, sw)
    p2 = ops.small_gemm_nt_split(sa, sw)
    assert torch.equal(p1, p2)
    assert relerr(p1, a.double() @ w.double().t()) < 2e-6


def test_small_gemm_unembed_store(dev):
    """HyperpriorDecoder un-embed fused into the GEMM store (vit_nlc.py:672-679): rearrange
    'b h w (p1 p2 c) -> b c (h p1) (w p2)' of Linear(360 -> 16*512) == the kernel's image output when the
    weight rows are pre-permuted to (c, p1, p2) order."""
    g = torch.Generator().manual_seed(9)
    Hz, Wz, p, cout, d = 18, 36, 4, 512, 360
    a = torch.randn(Hz * Wz, d, generator=g)
    w = torch.randn(p * p * cout, d, generator=g) * 0.05
    lin = (a.double() @ w.double().t()).view(Hz, Wz, p, p, cout)          # (h, w, p1, p2, c)
    ref = lin.permute(4, 0, 2, 1, 3).reshape(cout, Hz * p, Wz * p)
    wps = w.view(p, p, cout, d).permute(2, 0, 1, 3).reshape(p * p * cout, d).contiguous()
    img = torch.empty(cout, Hz * p, Wz * p, device=dev)
    ops.small_gemm_nt_split(ops.split_f16(a.to(dev)), ops.split_f16(wps.to(dev), "auto"), out=img,
                            unembed=(Hz, Wz, p, p))
    assert relerr(img, ref) < 2e-6
    # and it equals the un-fused route (row-major GEMM + pixel_shuffle kernel) to fp32 rounding
    lin32 = ops.small_gemm_nt_split(ops.split_f16(a.to(dev)), ops.split_f16(w.to(dev), "auto"))
    img2 = ops.pixel_shuffle(lin32, Hz, Wz, p, p)
    assert relerr(img, img2.double()) < 1e-6


@pytest.mark.parametrize("n,heads,hd", [(648, 5, 72), (648, 2, 72), (100, 3, 72), (41, 2, 64), (17, 1, 72)])
def test_hyper_attention_matches_fp64(dev, n, heads, hd):
    """Key-split exact-fp32 attention (csrc/hyper.hip) vs float64 softmax attention and vs the round-1
    window_attention_f32 kernel; ragged last query / key tiles; softmax spike rows."""
    g = torch.Generator().manual_seed(n + heads)
    C = heads * hd
    qkv = torch.randn(n, 3 * C, generator=g)
    qkv[3, :C] *= 6.0                       # a peaked row
    q, k, v = qkv.double().view(n, 3, heads, hd).permute(1, 2, 0, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).permute(1, 0, 2).reshape(n, C)
    dq = qkv.to(dev)
    so = ops.SplitMat.empty(n, C, dev, zero=True)
    out = ops.hyper_attention(dq, heads, out=torch.empty(n, C, device=dev), out_split=so)
    e = relerr(out, ref)
    print(f"hyper attention n={n} heads={heads} hd={hd}: rel {e:.2e}")
    assert e < 2e-6
    assert float((so.to_float() - out).abs().max()) <= 2 ** -21 * float(out.abs().max()) + 2 ** -24
    assert torch.equal(out, ops.hyper_attention(dq, heads))          # deterministic
    if hd == 72 and n == 648:
        old = ops.window_attention(dq, torch.zeros(3 * C, device=dev), heads, 18, 36, 18, 36)
        assert relerr(out, old.double()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(10368, 4096, 1024), (10368, 1024, 4096), (10368, 1024, 1024), (10368, 3072, 1024),
                                   (10368, 2048, 512), (4000, 1280, 2048), (10368, 1024, 7392)])
def test_gemm_model_shapes_all_epilogues(dev, M, N, K):
    """The big-tile split-f16 GEMM at the model's shapes with the full epilogue (bias + GELU + residual, fp32 and split
    outputs at once): fp32-accurate against float64, bit-reproducible from run to run."""
    g = torch.Generator().manual_seed(M + 7 * N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    sa, sw = ops.split_f16(a.to(dev)), ops.split_f16(w.to(dev), "auto")
    bd, rd = b.to(dev), r.to(dev)
    os_ = ops.SplitMat.empty(M, N, dev, zero=True)
    o1 = ops.gemm_nt_split(sa, sw, bias=bd, res=rd, gelu=True, out_split=os_)
    o2 = ops.gemm_nt_split(sa, sw, bias=bd, res=rd, gelu=True)
    torch.cuda.synchronize()
    e = relerr(o1, ref)
    print(f"gemm {M}x{N}x{K}: rel err {e:.2e}")
    assert torch.equal(o1, o2) and e < 2e-6
    assert float((os_.to_float() - o1).abs().max()) <= 2 ** -21 * float(o1.abs().max()) + 2 ** -24
    p2 = ops.gemm_nt_split(sa, sw)
    assert relerr(p2, a.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("C,K,denorm,hi", [(8, 128, True, False), (8, 128, False, False), (268, 1024, True, False),
                                            (5, 96, True, True), (159, 1024, True, False), (8, 1024, True, True)])
def test_fused_unembed_equals_gemm_plus_overlap_add(dev, C, K, denorm, hi):
    """cra5_gemm_nt_split_unembed (GEMM epilogue -> reconstruction, overlap rows through the side buffer + fix-up
    kernel; vit_nlc.py:628-630, 666-669) against the two-call form it replaces - GEMM into the [tokens][C*110] column
    matrix, then cra5_col2im_f32 - on the model's 72 x 144 token grid: BIT-identical images (same accumulators, same
    order of the two overlap contributions, same de-normalisation arithmetic), incl. row 0 / row 720 (one contribution),
    the ragged last tile column (C*110 is not a multiple of 256) and the half-filled last tile row (10 368 = 40.5 x 256)."""
    H, W, kh, kw, sh, sw = 721, 1440, 11, 10, 10, 10
    Hp, Wp = 72, 144
    g = torch.Generator().manual_seed(C * 7 + K)
    a = ops.split_f16(torch.randn(Hp * Wp, K, generator=g).to(dev))
    w = ops.split_f16((torch.randn(C * kh * kw, K, generator=g) / np.sqrt(K)).to(dev), "auto")
    mean = (torch.randn(C, generator=g) * 50).to(dev) if denorm else None
    std = (torch.rand(C, generator=g) * 30 + 0.5).to(dev) if denorm else None
    cols = ops.gemm_nt_split(a, w, hi_only=hi)
    ref = torch.full((C, H, W), float("nan"), device=dev)
    ops.col2im(cols, C, kh, kw, sh, sw, Hp, Wp, mean=mean, std=std, out=ref)
    nb = ops.unembed_side_bytes(C, H, W, kh, kw, sh, sw)
    assert nb == C * Hp * 2 * W * 4
    side = torch.full((nb // 4,), float("nan"), device=dev)
    out = torch.full((C, H, W), float("nan"), device=dev)
    ops.gemm_unembed(a, w, C, H, W, kh, kw, sh, sw, side, mean=mean, std=std, out=out, hi_only=hi)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()                       # every pixel written
    assert torch.equal(out, ref)
    # other geometries are refused (the caller keeps the two-call form), never mis-scattered
    assert ops.unembed_side_bytes(C, H, W, 4, 4, 4, 4) == 0
    with pytest.raises(Exception):
        ops.gemm_unembed(a, w, C, H, W, kh, kw, sh, sw, side[: nb // 8], out=out)


def test_fused_unembed_in_the_model_equals_the_two_call_form(dev):
    from cra5_amd.vaeformer import VAEformer
    net = VAEformer(0, **synth.thin_model_kwargs())
    synth.load_synthetic(net, seed=7)
    net = net.to(dev)
    y_hat = torch.round(2.0 * torch.randn(16, 72, 144, generator=torch.Generator().manual_seed(9))).to(dev)
    mean, std = torch.linspace(-3, 3, 8, device=dev), torch.linspace(0.5, 4, 8, device=dev)
    net.fused_unembed = True
    a = net._decode_frame(y_hat, mean=mean, std=std)
    b = net._decode_frame(y_hat)
    net.fused_unembed = False
    a2 = net._decode_frame(y_hat, mean=mean, std=std)
    b2 = net._decode_frame(y_hat)
    torch.cuda.synchronize()
    assert torch.equal(a, a2) and torch.equal(b, b2)


def test_global_attention_balanced_schedule_random_shapes(dev):
    """Randomised plans of the key-split schedule (token counts that leave 0 .. G*12 - 1 remainder tiles, partial last
    groups, 8 / 16 / 32 heads, n_full = 1 .. 3): against the plain launch - identical on the full-pass tokens, fp32 noise
    on the key-split ones, every token written."""
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    rng = np.random.default_rng(12)
    done = 0
    for _ in range(40):
        heads = int(rng.choice([8, 16, 32]))
        if cus % heads:
            continue
        tiles = int(rng.integers(12 * (cus // heads), 3 * 12 * (cus // heads) + 40))
        N, C = tiles * 32, 64 * heads
        if N * 3 * C * 4 > 1.5e9:
            continue
        ok, nb = ops.attention_balanced_plan(N, heads)
        if not ok:
            continue
        full_tiles, n_grp, maxp = _bal_plan(tiles, cus // heads)
        assert nb == heads * n_grp * maxp * 12 * 32 * 68 * 4
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        qkv = torch.randn(N, 3 * C, generator=g)
        qkv[:, :2 * C] *= 1.4
        qs = ops.split_f16(qkv.to(dev))
        pad = ops.split_f16(torch.zeros(1, 3 * C, device=dev))
        H, W = 32, tiles
        plain = ops.window_attention_split(qs, pad, heads, H, W, H, W, out=torch.empty(N, C, device=dev))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev).fill_(0xFF) if nb else None
        out = torch.full((N, C), float("nan"), device=dev)
        ops.window_attention_split(qs, pad, heads, H, W, H, W, out=out, workspace=ws, balanced=True)
        torch.cuda.synchronize()
        n_full = full_tiles * 32
        assert torch.isfinite(out).all(), (tiles, heads)
        assert torch.equal(out[:n_full], plain[:n_full]), (tiles, heads)
        if n_full < N:
            assert rmse(out[n_full:], plain[n_full:]) < 1e-6, (tiles, heads)
        done += 1
        del qkv, qs, plain, out, ws
        if done >= 12:
            break
    assert done >= 6


def test_window_attention_persistent_units_opt_in(dev):
    """Round 6 experiment kept as an opt-in flag of the C ABI (CRA5_ATTN_PERSISTENT_UNITS): windowed launches as persistent
    12-wave work-groups walking (window, head) units, the six left-over wave-tiles of a 576-token window run as a key-SPLIT
    unit whose two halves are merged through LDS.  Same bounds as the product path: fp32-accurate form against float64 at the
    exact-f32 kernel's accuracy class on the three window shapes (48 x 12: padded windows); tokens of FULL units are
    bit-identical to the default schedule (same arithmetic), the others differ by fp32 rounding; both reduced-precision
    layouts stay in their class."""
    H, W, C, heads = 72, 144, 128, 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, H * W, C, generator=g)
    sd = synth.fill_state_dict({"attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,), "attn.proj.weight": (C, C),
                                "attn.proj.bias": (C,)}, seed=21)
    sd["attn.qkv.weight"] *= 3.0
    xd = {k: v.double() for k, v in sd.items()}
    xs = ops.split_f16(x[0].to(dev))
    qkv_s = ops.SplitMat.empty(H * W, 3 * C, dev)
    ops.gemm_nt_split(xs, ops.split_f16(sd["attn.qkv.weight"].to(dev), "auto"), bias=sd["attn.qkv.bias"].to(dev),
                      out_split=qkv_s, want_f32=False)
    pad_s = ops.split_f16(sd["attn.qkv.bias"].to(dev).reshape(1, -1))
    for ws in ((24, 24), (12, 48), (48, 12)):
        ref = R.attention_window(x.double(), xd, "attn", heads, H, W, ws)
        att = ops.SplitMat.empty(H * W, C, dev, zero=True)
        o_p = torch.empty(H * W, C, device=dev)
        ops.window_attention_split(qkv_s, pad_s, heads, H, W, ws[0], ws[1], out=o_p, out_split=att, persistent_units=True)
        out = ops.gemm_nt_split(att, ops.split_f16(sd["attn.proj.weight"].to(dev), "auto"), bias=sd["attn.proj.bias"].to(dev))
        assert rmse(out, ref[0]) < 4e-6                      # the product path's bound (test_window_attention_split_f16)
        o_c = torch.empty(H * W, C, device=dev)
        ops.window_attention_split(qkv_s, pad_s, heads, H, W, ws[0], ws[1], out=o_c)
        same = float((o_p == o_c).float().mean())
        assert same >= 0.6 and float((o_p - o_c).abs().max()) < 1e-5, (ws, same)   # 12 of 18 wave-tiles per window run FULL units
        o16 = torch.empty(H * W, C, device=dev)
        ops.window_attention_split(qkv_s, pad_s, heads, H, W, ws[0], ws[1], out=o16, hi_only=True, persistent_units=True)
        assert torch.isfinite(o16).all() and rmse(o16, o_c) < 2e-3 * float(o_c.double().pow(2).mean().sqrt())
