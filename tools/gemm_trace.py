"""Per-block timeline of the split GEMM on the model's shapes (needs the trace variant):
   tools/build_variant.sh trace gemm_split_f16.hip -DCRA5_GEMM_TRACE
   CRA5_LIB=build_variants/libcra5_trace.so python tools/gemm_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cra5_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
L.cra5_debug_gemm_trace.restype = ctypes.c_int
L.cra5_debug_gemm_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
# (name, M, N, K, gelu, res, split_out)
SHAPES = [("qkv", 10368, 3072, 1024, False, False, True), ("proj", 10368, 1024, 1024, False, True, False),
          ("fc1", 10368, 4096, 1024, True, False, True), ("fc2", 10368, 1024, 4096, False, True, False)]
HI = "--hi" in sys.argv        # reduced-precision mode (CRA5_GEMM_HI_ONLY: the wide 64-k-step form)
if "--active" in sys.argv:
    # Power / clock experiment: ONE round of 256 x 256 tiles on 64 / 128 / 192 / 256 of the 256 CUs (N = 1024: 4 tile
    # columns; M = 256 r rows -> 4 r tiles), K = 8192 so that the round lasts ~0.6 ms.  If the chip is power-capped under
    # the dense MFMA stream, the shader clock falls as more CUs are active and (active CUs x clock) stays constant.
    os.environ["CRA5_GEMM_TILE"] = "256"
    SHAPES = [(f"act{4 * r}", 256 * r, 1024, 8192, False, False, False) for r in (8, 16, 32, 48, 64)]
if "--small" in sys.argv:   # a handful of tiles: prologue / epilogue without 256 CUs bursting together
    SHAPES = [("qkv16", 1024, 3072, 1024, False, False, True), ("fc1_16", 1024, 4096, 1024, True, False, True),
              ("proj8", 1536, 1024, 1024, False, True, False)]
for name, M, N, K, gelu, res, so in SHAPES:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    bias = torch.randn(N, device=dev)
    sa, sw = ops.split_f16(a), ops.split_f16(w, "auto")
    out = torch.empty(M, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    sm = ops.SplitMat.empty(M, N, dev) if so else None
    def run():
        if so:
            ops.gemm_nt_split(sa, sw, bias=bias, gelu=gelu, out_split=sm, want_f32=False, hi_only=HI)
        else:
            ops.gemm_nt_split(sa, sw, bias=bias, res=r, out=out, hi_only=HI)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    scratch = np.zeros((8192, 5), np.uint64)
    L.cra5_debug_gemm_trace(scratch.ctypes.data, 8192)   # clears the device buffer
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    nb = 8192
    buf = np.zeros((nb, 5), np.uint64)
    rc = L.cra5_debug_gemm_trace(buf.ctypes.data, nb)
    assert rc == 0
    t = buf[:, :4].astype(np.int64)
    used = t[:, 3] > 0
    t = t[used]
    n = len(t)
    # only the blocks of THIS launch: entries newer than the launch's first entry
    base = t[:, 0].min()
    us = (t - base) / 100.0     # 100 MHz wall clock -> us
    pro, main, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    span = us[:, 3].max()
    first = us[:, 0] < np.percentile(us[:, 0], 100.0 * min(1.0, 256.0 / n)) + 0.5
    print(f"{name:5s} {M}x{N}x{K}: event {e0.elapsed_time(e1)*1e3:7.1f} us, device span {span:7.1f} us, blocks {n}")
    print(f"      per block: prologue {pro.mean():5.1f}  main {main.mean():6.1f} (min {main.min():6.1f} max {main.max():6.1f})  "
          f"epilogue {epi.mean():5.1f} (max {epi.max():5.1f})  total {(us[:,3]-us[:,0]).mean():6.1f}")
    order = np.argsort(us[:, 0])
    starts = us[order, 0]
    print(f"      block start times: p0 {starts[0]:.1f} p25 {np.percentile(starts,25):.1f} p50 {np.percentile(starts,50):.1f} "
          f"p75 {np.percentile(starts,75):.1f} p100 {starts[-1]:.1f};  end p50 {np.percentile(us[:,3],50):.1f} p100 {span:.1f}")
    cyc = buf[:, 4].astype(np.int64)[used]
    ghz = cyc / ((t[:, 3] - t[:, 0]) * 10.0)     # shader cycles per ns
    print(f"      shader clock while the block ran: mean {ghz.mean():.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
    flop = 2.0 * M * N * K
    print(f"      rate: whole {flop/span/1e6:6.1f} TF, main-loop-only per block {flop/n/ (main.mean())/1e6*256:6.1f} TF-equivalent at 256 CUs")
    del a, w, sa, sw, out, r, sm
