"""Micro-benchmark of the GEMM engines on the model's shapes (GPU box).
   python tools/gemm_bench.py            (CRA5_GEMM_TILE=64|128|192|256 forces a tile config)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import ops

dev = torch.device("cuda:0")
SHAPES = [("qkv", 10368, 3072, 1024), ("proj", 10368, 1024, 1024), ("fc1", 10368, 4096, 1024),
          ("fc2", 10368, 1024, 4096), ("unembed", 10368, 29480, 1024), ("patch", 10368, 1024, 29480)]
only = sys.argv[1:] 
for name, M, N, K in SHAPES:
    if only and name not in only:
        continue
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    sa, sw = ops.split_f16(a), ops.split_f16(w, "auto")
    out = torch.empty(M, N, device=dev)
    for _ in range(2):
        ops.gemm_nt_split(sa, sw, out=out)
    torch.cuda.synchronize()
    n = 10 if M * N * K < 1e11 else 3
    t0 = time.perf_counter()
    for _ in range(n):
        ops.gemm_nt_split(sa, sw, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:8s} {M}x{N}x{K}: {dt*1e3:8.3f} ms  {2*M*N*K/dt/1e12:7.1f} TF (tile {os.environ.get('CRA5_GEMM_TILE','auto')})", flush=True)
    del a, w, sa, sw, out
