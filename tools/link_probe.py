"""Host link probe, free of the product's code: what does pinned host memory <-> HBM do on THIS box?

MEASUREMENT TOOL ONLY.  Talks to libamdhip64 through ctypes - no torch, no cra5_amd: nothing of the product's staging
code (cra5_copy_*_staged, the API's pinned pools, its copy threads) is between the numbers and the platform.

For the GPU's own NUMA node and for every other node of the box:
  * the calling thread (and the memory policy of its allocations) is bound to that node's CPUs,
  * two 1.11 GB (= one ERA5 frame) pinned buffers are allocated with hipHostMallocNumaUser (the pages follow the
    thread's node: checked against /proc/self/numa_maps) and touched,
  * H2D alone, D2H alone, and both at once on two streams are timed (wall clock around stream syncs, REPS frame-sized
    copies each), GB/s = 1e9 bytes/s.
Also: hipHostMalloc with DEFAULT flags (where do its pages land?), and the pageable -> pinned host memcpy the API's
batch path does per frame (numpy copyto, 1 / 4 / 8 threads bound to the node).

  python tools/link_probe.py [out.json]
"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

FRAME = 268 * 721 * 1440 * 4          # bytes of one fp32 ERA5 frame
REPS = 8
H2D, D2H = 1, 2
NUMA_USER = 0x20000000                # hipHostMallocNumaUser
NONBLOCK = 1                          # hipStreamNonBlocking

hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc} {hip.hipGetErrorString(rc).decode()}")


def cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def nodes():
    root = "/sys/devices/system/node"
    out = {}
    for d in sorted(os.listdir(root)):
        if d.startswith("node") and d[4:].isdigit():
            cpus = [c for c in cpulist(open(f"{root}/{d}/cpulist").read()) if c in os.sched_getaffinity(0)]
            if cpus:
                out[int(d[4:])] = cpus
    return out


def gpu_node():
    buf = ctypes.create_string_buffer(64)
    ck(hip.hipDeviceGetPCIBusId(buf, 64, 0), "hipDeviceGetPCIBusId")
    bdf = buf.value.decode().lower()
    try:
        return bdf, int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except (OSError, ValueError):
        return bdf, None


def node_of_address(addr):
    """NUMA node holding most pages of the mapping that starts at (or contains) addr, from /proc/self/numa_maps."""
    best = None
    try:
        for line in open("/proc/self/numa_maps"):
            parts = line.split()
            a = int(parts[0], 16)
            if a <= addr and (best is None or a > best[0]):
                counts = {int(p[1:].split("=")[0]): int(p.split("=")[1]) for p in parts[1:] if p[0] == "N" and "=" in p}
                best = (a, counts)
    except (OSError, ValueError):
        return None
    return best[1] if best else None


def host_alloc(nbytes, flags):
    p = ctypes.c_void_p()
    ck(hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(flags)), "hipHostMalloc")
    arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
    arr[::4096] = 1                    # touch every page from the bound thread
    return p, arr


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    ck(hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)), "hipMalloc")
    return p


def stream():
    s = ctypes.c_void_p()
    ck(hip.hipStreamCreateWithFlags(ctypes.byref(s), NONBLOCK), "hipStreamCreate")
    return s


def copies(dst, src, kind, s, reps=REPS):
    for _ in range(reps):
        ck(hip.hipMemcpyAsync(dst, src, ctypes.c_size_t(FRAME), kind, s), "hipMemcpyAsync")


def sync(s):
    ck(hip.hipStreamSynchronize(s), "hipStreamSynchronize")


def gbs(nbytes, dt):
    return nbytes / dt / 1e9


def link_legs(h_in, h_out, d_a, d_b, s0, s1):
    out = {}
    copies(d_a, h_in, H2D, s0, 2); copies(h_out, d_b, D2H, s1, 2); sync(s0); sync(s1)     # warm-up
    t = time.perf_counter(); copies(d_a, h_in, H2D, s0); sync(s0)
    out["h2d_alone_GBs"] = gbs(REPS * FRAME, time.perf_counter() - t)
    t = time.perf_counter(); copies(h_out, d_b, D2H, s1); sync(s1)
    out["d2h_alone_GBs"] = gbs(REPS * FRAME, time.perf_counter() - t)
    # both at once: the two streams are fed alternately, each direction's own finish time is taken by its own thread
    done = {}

    def waiter(name, s, t0):
        sync(s)
        done[name] = time.perf_counter() - t0

    t0 = time.perf_counter()
    for _ in range(REPS):
        copies(d_a, h_in, H2D, s0, 1)
        copies(h_out, d_b, D2H, s1, 1)
    th = [threading.Thread(target=waiter, args=("h2d", s0, t0)), threading.Thread(target=waiter, args=("d2h", s1, t0))]
    [x.start() for x in th]; [x.join() for x in th]
    wall = max(done.values())
    out["both_h2d_GBs"] = gbs(REPS * FRAME, done["h2d"])
    out["both_d2h_GBs"] = gbs(REPS * FRAME, done["d2h"])
    out["both_sum_GBs"] = gbs(2 * REPS * FRAME, wall)
    out["directions_add"] = out["both_sum_GBs"] / max(out["h2d_alone_GBs"], out["d2h_alone_GBs"])
    return out


def host_memcpy_leg(dst_arr, cpus, n_threads):
    """pageable -> pinned copy of one frame by n_threads threads bound to `cpus` (what the batch API does per frame)."""
    src = np.ones(FRAME, dtype=np.uint8)
    cut = [(i * FRAME // n_threads, (i + 1) * FRAME // n_threads) for i in range(n_threads)]

    def work(lo, hi):
        os.sched_setaffinity(0, cpus)
        np.copyto(dst_arr[lo:hi], src[lo:hi])

    best = 0.0
    for _ in range(3):
        th = [threading.Thread(target=work, args=c) for c in cut]
        t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]
        best = max(best, gbs(FRAME, time.perf_counter() - t))
    return best


def main():
    ck(hip.hipSetDevice(0), "hipSetDevice")
    bdf, gnode = gpu_node()
    nd = nodes()
    all_cpus = sorted(os.sched_getaffinity(0))
    out = {"gpu_bdf": bdf, "gpu_numa_node": gnode, "frame_bytes": FRAME, "reps": REPS,
           "nodes": {k: [len(v), v[0], v[-1]] for k, v in nd.items()}, "allowed_cpus": len(all_cpus), "legs": {}}
    print(f"GPU {bdf} on NUMA node {gnode}; nodes with allowed CPUs: {out['nodes']}", flush=True)
    d_a, d_b = dev_alloc(FRAME), dev_alloc(FRAME)
    s0, s1 = stream(), stream()

    # 1. default-flag pinned memory, thread unbound: where does the runtime put it, how fast is it?
    p_in, a_in = host_alloc(FRAME, 0)
    p_out, a_out = host_alloc(FRAME, 0)
    leg = link_legs(p_in, p_out, d_a, d_b, s0, s1)
    leg["pages_on_node"] = node_of_address(p_in.value)
    out["legs"]["default_flags_unbound_thread"] = leg
    print("default hipHostMalloc, unbound thread:", json.dumps(leg), flush=True)
    hip.hipHostFree(p_in); hip.hipHostFree(p_out)

    # 2. NumaUser pinned memory with the thread bound to each node in turn
    order = sorted(nd, key=lambda n: (n != gnode, n))
    for n in order:
        os.sched_setaffinity(0, nd[n])
        p_in, a_in = host_alloc(FRAME, NUMA_USER)
        p_out, a_out = host_alloc(FRAME, NUMA_USER)
        leg = link_legs(p_in, p_out, d_a, d_b, s0, s1)
        leg["pages_on_node"] = node_of_address(p_in.value)
        leg["host_memcpy_pageable_to_pinned_GBs"] = {str(k): host_memcpy_leg(a_in, nd[n], k) for k in (1, 4, 8)}
        tag = f"node{n}_{'gpu_local' if n == gnode else 'remote'}"
        out["legs"][tag] = leg
        print(f"NumaUser pinned, thread + memory on node {n} ({'GPU-local' if n == gnode else 'remote'}):",
              json.dumps(leg), flush=True)
        hip.hipHostFree(p_in); hip.hipHostFree(p_out)
        os.sched_setaffinity(0, all_cpus)

    loc = out["legs"].get(f"node{gnode}_gpu_local") or out["legs"]["default_flags_unbound_thread"]
    out["summary"] = {
        "one_direction_cap_GBs": max(loc["h2d_alone_GBs"], loc["d2h_alone_GBs"]),
        "both_directions_sum_GBs": loc["both_sum_GBs"],
        "directions_add": loc["directions_add"],
        "frames_per_s_h2d_bound": loc["h2d_alone_GBs"] * 1e9 / FRAME,
        "frames_per_s_d2h_bound": loc["d2h_alone_GBs"] * 1e9 / FRAME,
        "frames_per_s_round_trip_bound": loc["both_sum_GBs"] * 1e9 / (2 * FRAME),
    }
    print("summary:", json.dumps(out["summary"]), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
