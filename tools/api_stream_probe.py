"""What bounds the PCIe-inclusive batch API (bench.py `api_pipelined`)?  roundtrip_batch / encode_era5_batch / decode_batch
of the 268 model on host fp32 frames under a few configurations: frames in flight, host copy threads per frame.
  python tools/api_stream_probe.py [n_frames=36]"""
import os, sys, tempfile, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cra5_amd import synth
from cra5_amd.api import cra5_api
from cra5_amd.zoo import vaeformer_pretrained

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
net = vaeformer_pretrained(268); synth.load_synthetic(net, seed=7); net = net.to("cuda")
net.gpu_exclusive = False
tmp = tempfile.mkdtemp()
api = cra5_api(local_root=tmp, device="cuda", weights=net)
host = [(synth.synth_frame(268, seed=5 + i) * api.std.cpu() + api.mean.cpu()).numpy() for i in range(4)]
stamps = [f"2024-06-{1 + i // 24:02d}T{i % 24:02d}:00:00" for i in range(n)]
data = [host[i % 4] for i in range(n)]
tls = threading.local()


def consumer(i, arr):
    dst = getattr(tls, "dst", None)
    if dst is None:
        dst = tls.dst = np.empty(arr.shape, np.float32)
    np.copyto(dst, arr)
    return 0


for workers, ct in ((12, 1), (24, 1), (12, 4), (16, 4), (12, 8)):
    api.batch_copy_threads = ct
    out = np.empty((workers,) + host[0].shape, np.float32) if ct > 1 else None
    for rep in range(2):
        nn = n if rep else workers
        t0 = time.perf_counter()
        enc = api.encode_era5_batch(stamps[:nn], data=data[:nn], save_root=tmp + "/E", workers=workers)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        api.decode_batch(paths=[e["save_path"] for e in enc], workers=workers, sink=consumer)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        api.roundtrip_batch(stamps[:nn], data=data[:nn], save_root=tmp + "/R", workers=workers, sink=consumer)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"workers {workers:2d} copy threads {ct}: encode {nn / (t1 - t0):5.1f}  decode {nn / (t2 - t1):5.1f}  streamed round trip "
          f"{nn / (t3 - t2):5.1f} frames/s", flush=True)
