"""Kernel durations of the small-M GEMM instantiations per hyper-prior shape.  Run under rocprofv3:
   rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o hy -- python tools/hyper_gemm_sweep.py <TN> <KS>
(one process per instantiation: the override is read once), then the per-kernel averages of /tmp/p."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    os.environ["CRA5_HY_GEMM"] = f"{sys.argv[1]} {sys.argv[2]}"
import torch  # noqa: E402

from cra5_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
big = len(sys.argv) > 1 and sys.argv[1] == "big"
for (M, N, K) in ((648, 1080, 360), (648, 360, 360), (648, 1440, 360), (648, 360, 1440), (648, 360, 4096),
                  (648, 8192, 360), (648, 256, 360)):
    a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
    w = ops.split_f16((torch.randn(N, K, generator=g) * 0.05).to(dev), "auto")
    out = torch.empty(M, N, device=dev)
    marker = torch.zeros(N, device=dev)     # a fill kernel between shapes: separates the groups in the trace
    for _ in range(30):
        if big:
            ops.gemm_nt_split(a, w, out=out)
        else:
            ops.small_gemm_nt_split(a, w, out=out)
    torch.cuda.synchronize()
