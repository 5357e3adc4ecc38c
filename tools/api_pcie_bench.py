"""PCIe-inclusive rate through the cra5_api surface: host numpy frame in, host array out (GPU box)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cra5_amd import synth
from cra5_amd.api import cra5_api
from cra5_amd.zoo import vaeformer_pretrained
net = vaeformer_pretrained(268); synth.load_synthetic(net, seed=7)
tmp = tempfile.mkdtemp()
api = cra5_api(local_root=tmp, device="cuda", weights=net)
frame = (synth.synth_frame(268, seed=5) * api.std.cpu() + api.mean.cpu()).numpy()   # physical units, host
ts = "2024-06-01T00:00:00"
for i in range(4):
    t0 = time.perf_counter()
    r = api.encode_era5_as_bin(ts, save_root=tmp + "/CRA5", data=frame)
    t1 = time.perf_counter()
    d = api.decode_from_bin(ts, return_format="de_normalized")
    xh = d["x_hat"].cpu().numpy()
    t2 = time.perf_counter()
    print(f"iter {i}: encode (H2D 1.11 GB + g_a + rANS + .bin write) {t1-t0:.3f}s, decode (.bin read + rANS + g_s + D2H 1.11 GB) {t2-t1:.3f}s"
          f" -> {1/(t2-t0):.2f} frames/s serial, PCIe-inclusive; bin {os.path.getsize(r['save_path'])/1e6:.2f} MB", flush=True)
err = np.sqrt(np.mean(((xh - frame) / api.std.cpu().numpy()) ** 2))
print("normalised reconstruction RMSE of the (random-weight) codec:", float(err))

# ---- the batch API: frames streamed through the pipeline with pinned staging ---------------------------------
n = int(os.environ.get("N_FRAMES", "24"))
stamps = [f"2024-06-{1 + i // 24:02d}T{i % 24:02d}:00:00" for i in range(n)]
frames = [frame] * n        # host, pageable, physical units
out = np.empty((n,) + frame.shape, dtype=np.float32)
for rep in range(2):
    t0 = time.perf_counter()
    res = api.encode_era5_batch(stamps, data=frames, save_root=tmp + "/CRA5")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rec = api.decode_batch(stamps, out=out)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"batch rep {rep}: {n} frames, 12 in flight: encode {n / (t1 - t0):.2f} frames/s, decode {n / (t2 - t1):.2f} frames/s, "
          f"round trip {n / (t2 - t0):.2f} frames/s (host array in -> .bin -> host array out, PCIe-inclusive)", flush=True)
err = np.sqrt(np.mean(((out[3] - frame) / api.std.cpu().numpy()) ** 2))
print("batch path: normalised reconstruction RMSE", float(err), "; equals the single-frame path:", bool(np.array_equal(out[3], xh)))
