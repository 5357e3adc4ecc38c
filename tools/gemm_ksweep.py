"""t(K) sweep of the split GEMM: fixed overhead (prologue + epilogue) vs main-loop rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import ops
dev = torch.device("cuda:0")
for (M, N) in ((10368, 3072), (10368, 1024), (10368, 4096)):
    res = []
    for K in (256, 512, 1024, 2048, 4096, 8192):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02
        sa, sw = ops.split_f16(a), ops.split_f16(w, "auto")
        out = torch.empty(M, N, device=dev)
        for _ in range(3): ops.gemm_nt_split(sa, sw, out=out)
        torch.cuda.synchronize(); n = 20
        t0 = time.perf_counter()
        for _ in range(n): ops.gemm_nt_split(sa, sw, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        res.append((K, dt))
        del a, w, sa, sw, out
    (k1, t1), (k2, t2) = res[2], res[-1]
    b = (t2 - t1) / (k2 - k1); a0 = t1 - b * k1
    print(f"M={M} N={N} tile={os.environ.get('CRA5_GEMM_TILE','auto')}: " + " ".join(f"K{k}:{t*1e6:.0f}us" for k, t in res) +
          f" | fixed {a0*1e6:.0f} us, main loop {2*M*N/b/1e12:.0f} TF", flush=True)
