"""Where does the host time between GPU phases go?  (GPU box; single frame, no concurrency)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cra5_amd import synth
from cra5_amd.zoo import vaeformer_pretrained
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False); synth.load_synthetic(net, seed=7); net = net.to(dev)
x = synth.synth_frame(268, seed=2).unsqueeze(0).to(dev)
out = net.compress(x); net.decompress(out["strings"], out["z_shape"])
eb, gc = net.entropy_bottleneck, net.gaussian_conditional
T = time.perf_counter
for rep in range(2):
    with net._gpu_phase():
        y = net._encode_y_frame(x[0]); s = net._latent_side_frame(y.contiguous())
        z_sym = net._to_host("z_sym", s["z_sym"]); y_sym = net._to_host("y_sym", s["y_sym"]); idx = net._to_host("idx", s["idx"])
    t0 = T(); z_idx = eb._build_indexes((1, z_sym.shape[0], z_sym.shape[1]))
    t1 = T(); z_str = eb.encode_symbols(z_sym.numpy().reshape(-1), z_idx)
    t2 = T(); ys = y_sym.numpy().reshape(-1); ii = idx.numpy().reshape(-1)
    t3 = T(); y_str = gc.encode_symbols(ys, ii)
    t4 = T()
    print(f"H1: build z idx {1e3*(t1-t0):.2f}  encode z {1e3*(t2-t1):.2f}  numpy views {1e3*(t3-t2):.2f}  encode y {1e3*(t4-t3):.2f} ms")
    Cz = eb.channels; zh, zw = net.Hz, net.Wz
    t0 = T(); z_idx = eb._build_indexes((1, Cz, zh, zw))
    t1 = T(); z_host = net._pinned("z_in", (Cz, zh * zw), torch.int32)
    t2 = T(); eb.decode_symbols(z_str, z_idx, out=z_host.numpy().reshape(-1))
    t3 = T()
    print(f"H1b: build z idx {1e3*(t1-t0):.2f}  pinned {1e3*(t2-t1):.2f}  decode z {1e3*(t3-t2):.2f} ms")
    y_host = net._pinned("y_in", (256, 72, 144), torch.int32)
    t0 = T(); a = idx.numpy().reshape(-1); o = y_host.numpy().reshape(-1)
    t1 = T(); gc.decode_symbols(y_str, a, out=o)
    t2 = T()
    print(f"H2: views {1e3*(t1-t0):.2f}  decode y {1e3*(t2-t1):.2f} ms  (y bytes {len(y_str)})")
    # raw coder on the same data for reference
    cdf, ln, off = gc.host_tables()
    from cra5_amd import ops
    t0 = T(); ops.rans_decode(y_str, a, cdf, ln, off, out=o); t1 = T(); ops.rans_encode(ys, ii, cdf, ln, off); t2 = T()
    print(f"raw ops: decode {1e3*(t1-t0):.2f}  encode {1e3*(t2-t1):.2f} ms;  idx histogram top: {np.bincount(ii).argsort()[-5:][::-1]}  escapes: {(np.abs(ys) > 1000).sum()}")
if os.environ.get("CRA5_DUMP_SYMBOLS"):
    np.savez_compressed(os.environ["CRA5_DUMP_SYMBOLS"], y_sym=ys.astype(np.int16) if np.abs(ys).max() < 32767 else ys, idx=ii.astype(np.uint8))
    print("dumped", os.environ["CRA5_DUMP_SYMBOLS"])
