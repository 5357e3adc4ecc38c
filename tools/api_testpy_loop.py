"""The reference's test.py loop (test.py:14-45: encode_to_latent, latent_to_bin, encode_era5_as_bin, bin_to_latent,
latent_to_reconstruction, decode_from_bin x 2 per iteration) on an in-memory host frame, single-threaded, on the
GPU box; prints per-call times and the serial single-frame rate a drop-in caller sees:

    frames/s = 1 / (encode_era5_as_bin(host array -> .bin on disk) + decode_from_bin(.bin -> x_hat))

with x_hat left on the device like the reference does, and with x_hat brought to a host array (to_host=True)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cra5_amd import synth  # noqa: E402
from cra5_amd.api import cra5_api  # noqa: E402
from cra5_amd.zoo import vaeformer_pretrained  # noqa: E402


def run(api, frame, n_iter, root):
    ts = "2024-06-01T00:00:00"
    t = {k: [] for k in ("encode_to_latent", "latent_to_bin", "encode_era5_as_bin", "bin_to_latent",
                         "latent_to_reconstruction", "decode_normalized", "decode_de_normalized", "decode_to_host")}
    host_out = np.empty(frame.shape, dtype=np.float32)

    def timed(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        t[key].append(time.perf_counter() - t0)
        return r
    for _ in range(n_iter):
        y = timed("encode_to_latent", lambda: api.encode_to_latent(time_stamp=ts, data=frame))
        timed("latent_to_bin", lambda: api.latent_to_bin(y=y))
        out = timed("encode_era5_as_bin", lambda: api.encode_era5_as_bin(time_stamp=ts, save_root=root + "/CRA5", data=frame))
        y_hat = timed("bin_to_latent", lambda: api.bin_to_latent(bin_path=out["save_path"]))
        timed("latent_to_reconstruction", lambda: api.latent_to_reconstruction(y_hat=y_hat))
        timed("decode_normalized", lambda: api.decode_from_bin(ts, return_format='normalized'))
        timed("decode_de_normalized", lambda: api.decode_from_bin(ts, return_format='de_normalized'))
        timed("decode_to_host", lambda: api.decode_from_bin(ts, return_format='de_normalized', out=host_out))
    med = {k: float(np.median(v[1:] or v)) for k, v in t.items()}
    return med


def main():
    n_iter = int(os.environ.get("N_ITER", "5"))
    net = vaeformer_pretrained(268)
    synth.load_synthetic(net, seed=7)
    root = tempfile.mkdtemp()
    api = cra5_api(local_root=root, device="cuda", weights=net)
    frame = (synth.synth_frame(268, seed=5) * api.std.cpu() + api.mean.cpu()).numpy()     # host, pageable, physical units
    med = run(api, frame, n_iter, root)
    res = dict(median_s=med,
               single_frame_fps_device_out=1.0 / (med["encode_era5_as_bin"] + med["decode_de_normalized"]),
               single_frame_fps_host_out=1.0 / (med["encode_era5_as_bin"] + med["decode_to_host"]),
               testpy_iteration_s=sum(v for k, v in med.items() if k != "decode_to_host"))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
