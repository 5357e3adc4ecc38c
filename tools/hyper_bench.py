"""Device time of the hyper-prior path (h_a, h_s) and of its kernels, HIP events, 268 model.
    python tools/hyper_bench.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cra5_amd import ops, synth  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (time.perf_counter() - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    net = vaeformer_pretrained(quality=268, pretrained=False)
    synth.load_synthetic(net, seed=7)
    net = net.to(dev)
    y = torch.randn(256, 72, 144, device=dev)
    z = net._h_a_frame(y)
    zh = torch.round(z)
    print("h_a  : %.1f us device, %.1f us wall" % timed(lambda: net._h_a_frame(y)))
    print("h_s  : %.1f us device, %.1f us wall" % timed(lambda: net._h_s_frame(zh)))
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in ((648, 1080, 360), (648, 360, 360), (648, 1440, 360), (648, 360, 1440), (648, 360, 4096),
                      (648, 8192, 360)):
        a = ops.split_f16(torch.randn(M, K, generator=g).to(dev))
        w = ops.split_f16((torch.randn(N, K, generator=g) * 0.05).to(dev), "auto")
        out = torch.empty(M, N, device=dev)
        ts = timed(lambda: ops.small_gemm_nt_split(a, w, out=out), 50)[0]
        tb = timed(lambda: ops.gemm_nt_split(a, w, out=out), 50)[0]
        print(f"gemm {M}x{N}x{K}: small {ts:.1f} us, big-engine {tb:.1f} us")
    qkv = torch.randn(648, 1080, generator=g).to(dev)
    o = torch.empty(648, 360, device=dev)
    print("attention 648x5x72: new %.1f us, old %.1f us" % (
        timed(lambda: ops.hyper_attention(qkv, 5, out=o), 50)[0],
        timed(lambda: ops.window_attention(qkv, torch.zeros(1080, device=dev), 5, 18, 36, 18, 36, out=o), 50)[0]))
    x = torch.randn(648, 360, device=dev)
    sm = ops.SplitMat.empty(648, 360, dev)
    ga = torch.ones(360, device=dev)
    print("layernorm 648x360: %.1f us" % timed(lambda: ops.layernorm(x, ga, ga, out_split=sm, want_f32=False), 50)[0])


if __name__ == "__main__":
    main()
