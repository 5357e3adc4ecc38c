"""A/B of the stream-K schedule vs the plain launch per model shape (HIP events, un-overlapped)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ws = ops.gemm_sk_workspace(dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, (M, N, K), epi in (("qkv", (10368, 3072, 1024), "bias_split"), ("proj", (10368, 1024, 1024), "res"),
                             ("fc1", (10368, 4096, 1024), "gelu_split"), ("fc2", (10368, 1024, 4096), "res"),
                             ("pe chunk", (10368, 1024, 7392), "res"), ("unembed", (10368, 29480, 1024), "none")):
    a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
    w = ops.split_f16((torch.randn(N, K, generator=g) * 0.03).to(dev), "auto")
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N if epi == "res" else 1, generator=g).to(dev) if epi == "res" else None
    out = torch.empty(M, N, device=dev) if "split" not in epi else None
    osp = ops.SplitMat.empty(M, N, dev, zero=True) if "split" in epi else None
    kw = dict(bias=b if epi != "none" else None, res=r, gelu="gelu" in epi, out=out, out_split=osp,
              want_f32=out is not None)
    fns = (lambda: ops.gemm_nt_split(a, w, sk_ws=ws, sk=False, **kw),
           lambda: ops.gemm_nt_split(a, w, sk_ws=ws, sk=True, **kw),
           lambda: ops.gemm_nt_split(a, w, sk_ws=ws, **kw))
    best = [1e30, 1e30, 1e30]
    for rep in range(4):          # interleaved A/B/C: the clock / cache state drifts with what ran before
        for i, f in enumerate(fns):
            best[i] = min(best[i], timed(f, 20))
    t_plain, t_sk, t_auto = best
    fl = 2.0 * M * N * K
    print(f"{name:9s} {M}x{N}x{K}: plain {t_plain:7.1f} us ({fl / t_plain / 1e6:5.0f} TF/s)  stream-K {t_sk:7.1f} us "
          f"({fl / t_sk / 1e6:5.0f} TF/s)  auto {t_auto:7.1f} us")
