"""Per-shape timing of the split GEMM on the model's shapes (HIP events, un-overlapped).

  python tools/gemm_bench.py            best-of-4 x 20 launches per shape (A/B: run under different CRA5_LIB)
  python tools/gemm_bench.py --once     3 launches per shape, for a `rocprofv3 --pmc FETCH_SIZE` pass around it
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

once = "--once" in sys.argv
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = (("qkv", (10368, 3072, 1024), "bias_split"), ("proj", (10368, 1024, 1024), "res"),
          ("fc1", (10368, 4096, 1024), "gelu_split"), ("fc2", (10368, 1024, 4096), "res"),
          ("pe chunk", (10368, 1024, 7392), "res"), ("unembed", (10368, 29480, 1024), "none"))


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fns = []
for name, (M, N, K), epi in SHAPES:
    a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
    w = ops.split_f16((torch.randn(N, K, generator=g) * 0.03).to(dev), "auto")
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if epi == "res" else None
    out = torch.empty(M, N, device=dev) if "split" not in epi else None
    osp = ops.SplitMat.empty(M, N, dev, zero=True) if "split" in epi else None
    kw = dict(bias=b if epi != "none" else None, res=r, gelu="gelu" in epi, out=out, out_split=osp,
              want_f32=out is not None)
    fns.append((name, 2.0 * M * N * K, (lambda a=a, w=w, kw=kw: ops.gemm_nt_split(a, w, **kw))))
if once:
    for name, fl, f in fns:
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    sys.exit(0)
best = {name: 1e30 for name, _, _ in fns}
for rep in range(4):
    for name, fl, f in fns:
        best[name] = min(best[name], timed(f, 20))
print(os.environ.get("CRA5_LIB", "default"), " ".join(f"{name} {best[name]:.1f} us ({fl / best[name] / 1e6:.0f} TF)" for name, fl, _ in fns))
