"""Frame-level software pipeline: several frames in flight on one GPU.

A frame's round trip alternates GPU phases (g_a, h_s, g_s - tens of ms of MFMA work) with
serial host phases (rANS encode / decode, a few tens of ms on one core).  Frames are
independent, so `FramePipeline` runs W frames concurrently, each on
its own host thread + HIP stream + activation workspace; while one frame sits in the
entropy coder on a CPU core, another frame's kernels occupy the GPU.  The native calls
release the GIL (ctypes), the weights are shared read-only.
"""
import os
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import torch


class FramePipeline:
    def __init__(self, net, workers=3, device=None):
        self.net = net
        self.device = torch.device(device) if device is not None else net.device
        self.workers = max(1, int(workers))
        self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="cra5-frame")
        # A frame thread coming back from a GIL-free entropy-coder call must re-take the GIL from the
        # threads that are busy launching kernels; CPython hands it over only every switch interval
        # (5 ms by default - measured: +20 ms on each 13-18 ms host phase with 8 frames in flight).
        si = float(getattr(getattr(net, "runtime", None), "switch_interval_s", 0.0002))
        if si > 0 and sys.getswitchinterval() > si:
            sys.setswitchinterval(si)
        self._tls = threading.local()

    def _stream(self):
        s = getattr(self._tls, "stream", None)
        if s is None:
            torch.cuda.set_device(self.device)
            s = self._tls.stream = torch.cuda.Stream(device=self.device)
        return s

    def _run(self, fn, item):
        s = self._stream()
        with torch.cuda.stream(s):
            out = fn(item)
        s.synchronize()
        return out

    def map(self, fn, items):
        """Run fn(item) for every item, up to `workers` at a time, each on its own stream.
        Returns the results in order (blocks until all are done)."""
        torch.cuda.current_stream(self.device).synchronize()  # inputs produced on the caller's stream
        futs = [self._pool.submit(self._run, fn, it) for it in items]
        return [f.result() for f in futs]

    # ---- the reference API's operations, many frames at a time ------------------------
    def compress(self, frames):
        """frames: iterable of [1, C, H, W] device tensors -> list of compress() dicts."""
        return self.map(self.net.compress, frames)

    def decompress(self, outs, return_format="reconstructed"):
        return self.map(lambda o: self.net.decompress(o["strings"], o["z_shape"], return_format), outs)

    def roundtrip(self, frames):
        """x -> .bin strings -> x_hat for every frame. Returns [(compress_out, x_hat), ...]."""
        def rt(x):
            out = self.net.compress(x)
            rec = self.net.decompress(out["strings"], out["z_shape"])
            return out, rec["x_hat"]
        return self.map(rt, frames)

    def close(self):
        self._pool.shutdown(wait=True)
