"""Which engine moves a pinned D2H / H2D copy, and how fast (GPU box): times torch's non_blocking copies of 24 MB and
1.11 GB between device and pinned host memory; run it under `rocprofv3 --kernel-trace --stats` to see whether
`__amd_rocclr_copyBuffer` blit kernels (copies on the CUs) or the SDMA engines (no kernel) did the work, with and
without HSA_ENABLE_SDMA / GPU_FORCE_BLIT_COPY_SIZE in the environment."""
import os
import time

import torch

dev = torch.device("cuda:0")
for mb in (24, 1113):
    n = mb * (1 << 20) // 4
    d = torch.randn(n, device=dev)
    h = torch.empty(n, pin_memory=True)
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{os.environ.get('HSA_ENABLE_SDMA', 'unset')}/{os.environ.get('GPU_FORCE_BLIT_COPY_SIZE', 'unset')} {name} {mb} MB: {dt * 1e3:.2f} ms = {mb / 1024 / dt:.1f} GB/s", flush=True)
