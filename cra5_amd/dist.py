"""Multi-GPU driver pieces: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by FRAME: hourly snapshots are independent units
(no cross-frame state, one rANS stream pair per frame - entropy_models.py:263-272 in the
reference), so weights are replicated and there is NO data-path collective.  The single
exchange is an all-gather of a tiny per-frame stats tensor (bytes of the y / z streams,
escape count, stream checksum): 40 B per frame, latency-bound, so ring-vs-tree or xGMI
link bandwidth is irrelevant for it.

Host side of an N-rank job: `init_from_env` pins the rank to its GPU's NUMA share of the host cores BEFORE the
process group and the HIP runtime exist (their helper threads inherit the mask; threads that already run are
re-pinned one by one), finding the GPU's PCI address without a HIP call (KFD topology in sysfs, honouring
HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES).
"""
import os
import zlib

import torch
import torch.distributed as dist

STATS_FIELDS = ("frame", "y_bytes", "z_bytes", "crc32", "n_escape")   # SURVEY 8(e); n_escape = -1: not counted


LAST_BIND = None    # what bind_rank_to_numa did for this process (bench.py reports it)


def init_from_env(device_type="cuda", numa_bind=False, bind_single=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run). Returns (rank, world, local_rank).  numa_bind: pin the rank to its GPU's NUMA share of the
    host cores first - before init_process_group and before the first HIP call of this function, so that the RCCL
    proxy threads, the HIP runtime's threads and every later frame thread are created under the mask."""
    global LAST_BIND
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if numa_bind and (world > 1 or bind_single):     # bind_single: a 1-rank job binds to its GPU's node too (round 6)
        # ROCr re-pins its own helper threads (async-event loop) to ALL cpus unless told to inherit the creator's mask
        os.environ.setdefault("HSA_OVERRIDE_CPU_AFFINITY_DEBUG", "0")
        LAST_BIND = bind_rank_to_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    force = os.environ.get("CRA5_FORCE_DIST") == "1"   # exercise the RCCL path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if device_type == "cuda" else "gloo"
        if os.environ.get("CRA5_DIST_BACKEND"):          # tests / emergencies: force "gloo"
            backend = os.environ["CRA5_DIST_BACKEND"]
        kw = {}
        if backend == "nccl":
            dev_i = local
            if os.environ.get("CRA5_SHARE_GPU") == "1":    # tests on a 1-GPU box: ranks share the visible GPUs
                dev_i = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev_i)
            kw["device_id"] = torch.device("cuda", dev_i)   # binds the RCCL communicator to this rank's GPU (eager init)
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("CRA5_DIST_TIMEOUT_S", "600"))), **kw)
    return rank, world, local


def _coll_device(device):
    """Collectives run on the device for RCCL ("nccl") and on the host for gloo (CPU tests, CRA5_DIST_BACKEND=gloo)."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _visible_physical_index(local):
    """Physical (ROCr enumeration) index of logical device `local` under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES remapping (integer lists only; UUID forms -> None)."""
    idx = local
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):   # HIP's list indexes ROCr's
        v = os.environ.get(var)
        if v is None or v.strip() == "":
            continue
        if var == "CUDA_VISIBLE_DEVICES" and os.environ.get("HIP_VISIBLE_DEVICES"):
            continue                                      # the two are aliases: apply once
        try:
            lst = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None
        if idx >= len(lst):
            return None
        idx = lst[idx]
    return idx


def visible_gpu_count(n_physical):
    """Logical GPUs this process sees: every physical one, or the entries of the visibility lists that name one."""
    n = 0
    while n < 64:
        phys = _visible_physical_index(n)
        if phys is None or phys >= n_physical:
            break
        n += 1
    return n


def kfd_gpu_bdfs(root="/sys/class/kfd/kfd/topology/nodes"):
    """PCI addresses of the GPUs in KFD topology order (= ROCr / HIP enumeration order without visibility masks), read
    from sysfs: no HIP call, usable before the runtime is initialised.  [] when the topology is not readable."""
    out = []
    try:
        ids = sorted((d for d in os.listdir(root) if d.isdigit()), key=int)
    except OSError:
        return out
    for d in ids:
        props = {}
        try:
            for line in open(os.path.join(root, d, "properties")):
                k, _, v = line.strip().partition(" ")
                props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue                                      # a CPU node
        loc, dom = int(props.get("location_id", "0") or 0), int(props.get("domain", "0") or 0)
        out.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out


def gpu_numa_node(local, bdfs=None, sys_pci="/sys/bus/pci/devices"):
    """NUMA node of logical GPU `local` from its PCI address, or None when it cannot be told.  The address comes from
    the KFD topology + the visibility masks (no HIP call: this runs before the process group / HIP runtime start);
    torch's device properties are the fallback when HIP is already up."""
    bdf = None
    phys = _visible_physical_index(local)
    bdfs = kfd_gpu_bdfs() if bdfs is None else bdfs
    if phys is not None and phys < len(bdfs):
        bdf = bdfs[phys]
    elif torch.cuda.is_initialized():
        try:
            p = torch.cuda.get_device_properties(local)
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        except Exception:  # noqa: BLE001
            bdf = None
    if bdf is None:
        return None
    try:
        node = int(open(f"{sys_pci}/{bdf}/numa_node").read())
        return node if node >= 0 else None
    except (OSError, ValueError):
        return None


def rank_cpu_set(local, n_local, node=None, node_peers=None, allowed=None, node_cpus=None):
    """The host CPUs rank `local` of `n_local` ranks on this node should run its frame threads and rANS
    work on.  With a known NUMA node: that node's CPUs, cut into equal contiguous shares among the
    `node_peers` = (index of this rank among the ranks on the same node, how many there are).  Without:
    an equal contiguous share of the allowed CPUs.  Pure function of its arguments (tested on CPU)."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    cpus = allowed
    k, m = local, n_local
    if node is not None and node_cpus:
        on_node = [c for c in node_cpus if c in set(allowed)]
        if on_node:
            cpus = on_node
            k, m = node_peers if node_peers else (0, 1)
    m = max(1, m)
    if len(cpus) < m:
        return cpus
    # A node's list is "cores..., SMT siblings..." (EPYC: 0-63,128-191): take the k-th share of EVERY contiguous
    # run, so that a rank gets whole cores (a core and its sibling) instead of sharing cores with another rank
    runs, cur = [], [cpus[0]]
    for c in cpus[1:]:
        if c == cur[-1] + 1:
            cur.append(c)
        else:
            runs.append(cur)
            cur = [c]
    runs.append(cur)
    out = []
    for r in runs:
        share = len(r) // m
        out.extend(r[k * share:(k + 1) * share] if share else [])
    return out or cpus[k * (len(cpus) // m):(k + 1) * (len(cpus) // m)]


def plan_rank_cpus(local, n_local, nodes, allowed, node_cpus_of):
    """CPU set of rank `local`: pure function (tested on CPU).  nodes[i] = NUMA node of rank i's GPU or None.  If ANY
    rank's node is unknown, EVERY rank takes an equal share of the allowed CPUs - one consistent rule, so that a
    node-bound rank and a fallback rank can never be handed overlapping cores."""
    if any(n is None for n in nodes):
        return None, rank_cpu_set(local, n_local, None, None, allowed, None)
    node = nodes[local]
    same = [i for i in range(n_local) if nodes[i] == node]
    return node, rank_cpu_set(local, n_local, node, (same.index(local), len(same)), allowed, node_cpus_of(node))


def _pin_all_threads(cpus):
    """sched_setaffinity(0) moves only the calling thread: pin every thread this process already has (OpenMP pool,
    anything an import started); threads created later inherit the caller's mask."""
    n = 0
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    for t in tids:
        try:
            os.sched_setaffinity(t, cpus)
            n += 1
        except OSError:
            pass
    os.sched_setaffinity(0, cpus)
    return n


def bind_rank_to_numa(local, n_local):
    """Pin this process - the threads it has and all its future ones (the 12 frame threads, the rANS pool, RCCL's and
    the HIP runtime's helpers when called before they start) - to the CPUs of its GPU's NUMA node, shared equally
    with the other ranks whose GPU sits on the same node.  Returns a dict describing what was done (reported in
    bench.py's JSON line)."""
    info = {"numa_node": None, "cpus": None, "bound": False}
    try:
        bdfs = kfd_gpu_bdfs()
        n_vis = visible_gpu_count(len(bdfs))
        share = os.environ.get("CRA5_SHARE_GPU") == "1" and n_vis > 0     # tests: ranks share the visible GPUs
        nodes = [gpu_numa_node(i % n_vis if share else i, bdfs) for i in range(n_local)]
        node, cpus = plan_rank_cpus(
            local, n_local, nodes, None,
            lambda nd: _parse_cpulist(open(f"/sys/devices/system/node/node{nd}/cpulist").read()))
        if cpus:
            n_thr = _pin_all_threads(cpus)
            info.update(numa_node=node, cpus=len(cpus), bound=True, first_cpu=cpus[0], last_cpu=cpus[-1],
                        threads_pinned=n_thr, before_hip_init=not torch.cuda.is_initialized(),
                        rule="numa share" if node is not None else "equal share of the allowed cpus (a GPU's node unknown)")
    except Exception as e:  # noqa: BLE001
        info["error"] = repr(e)
    return info


def host_report():
    """Where this rank's threads may run: its own mask and how many of its threads have a mask outside it (must be
    0 after bind_rank_to_numa).  all_gather_object'ed into bench.py's JSON line for N > 1."""
    own = set(os.sched_getaffinity(0))
    outside = total = 0
    names = {}
    try:
        for t in os.listdir("/proc/self/task"):
            total += 1
            try:
                if not set(os.sched_getaffinity(int(t))) <= own:
                    outside += 1
                    try:
                        nm = open(f"/proc/self/task/{t}/comm").read().strip()
                    except OSError:
                        nm = "?"
                    m = sorted(os.sched_getaffinity(int(t)))
                    names[nm] = names.get(nm, 0) + 1
                    names[f"{nm}:mask"] = [len(m), m[0], m[-1]]
            except OSError:
                pass
    except OSError:
        pass
    cpus = sorted(own)
    return {"rank": int(os.environ.get("RANK", "0")), "pid": os.getpid(), "n_cpus": len(cpus), "cpus": cpus,
            "threads": total, "threads_outside_mask": outside, "outside_names": names, "gpu_check": gpu_bind_check()}


def gpu_bind_check(local=None):
    """AFTER the HIP runtime is up: does the PCI address the NUMA bind derived from the KFD topology + the visibility
    masks (before HIP existed) equal the address HIP reports for the device this rank actually drives?  The KFD node order
    is assumed to be ROCr's enumeration order and UUID-form visibility lists cannot be resolved from sysfs at all
    (ADVICE r4): a mismatch or an unresolved case is REPORTED in bench.py's JSON line (`config.host.per_rank[].gpu_check`)
    instead of being trusted silently.  {"kfd_bdf", "hip_bdf", "match": True | False | None}."""
    out = {"kfd_bdf": None, "hip_bdf": None, "match": None}
    try:
        if not torch.cuda.is_initialized():
            return out
        local = torch.cuda.current_device() if local is None else local
        p = torch.cuda.get_device_properties(local)
        out["hip_bdf"] = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        phys = _visible_physical_index(local)
        bdfs = kfd_gpu_bdfs()
        if phys is not None and phys < len(bdfs):
            out["kfd_bdf"] = bdfs[phys]
            out["match"] = out["kfd_bdf"][:-1] == out["hip_bdf"][:-1]        # (function digit aside)
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)
    return out


def gather_objects(obj):
    """all_gather_object over the job (a list with one entry per rank; [obj] without a process group)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def shard_frames(n_frames, rank, world):
    """Contiguous block partition: rank r owns frames [lo, hi). 64 frames on 8 GPUs ->
    [8r, 8r+8).  Remainders go to the lowest ranks."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return range(lo, hi)


def frame_stats(frame_id, strings, n_escape=-1):
    """One int64 row per frame: (frame, y_bytes, z_bytes, crc32 of y||z, escape symbols of the y stream)."""
    y, z = strings[0][0], strings[1][0]
    return [int(frame_id), len(y), len(z), zlib.crc32(z, zlib.crc32(y)) & 0xFFFFFFFF, int(n_escape)]


def gather_stats(rows, device):
    """all_gather of the per-rank stats (padded to the max per-rank frame count; -1 rows are
    padding).  Returns an int64 tensor [n_frames_total, 5] sorted by frame id on every rank."""
    device = _coll_device(device)
    t = torch.tensor(rows, dtype=torch.int64, device=device).reshape(-1, len(STATS_FIELDS))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t[torch.argsort(t[:, 0])] if t.numel() else t
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    m = int(max(int(v) for v in ns))
    pad = torch.full((m, len(STATS_FIELDS)), -1, dtype=torch.int64, device=device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    allr = torch.cat(out)
    allr = allr[allr[:, 0] >= 0]
    return allr[torch.argsort(allr[:, 0])]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


# ---- first-contact preflight of an N-rank job (round 6) ---------------------------------------------
# The first N = 8 run on real hardware is also the first time RCCL, the NUMA bind, 8 x 12 frame threads and 8 GPUs meet.
# Before the warm-up every rank reports what it found (a tiny collective of its own, free device memory, host CPUs in its
# mask, whether the GPU it drives is the one the NUMA bind assumed) and ALL ranks derive the same frames-in-flight figure
# from the gathered reports: a job that cannot keep the requested pipeline depth lowers it and says so in the JSON line
# instead of dying in the warm-up (out of memory) or crawling (12 frame threads on 4 cores).

GIB = float(1 << 30)
# device memory of the frame pipeline (soak: 61.8 GiB with 12 in flight and a 24-frame pool):
FRAME_BYTES = 268 * 721 * 1440 * 4                 # one fp32 ERA5 frame, 1.11 GB
BASE_BYTES = int(3.6 * GIB)                        # weights fp32 + split-f16 copies + tables + slack
PER_INFLIGHT_BYTES = int(2.2 * GIB) + FRAME_BYTES  # a frame thread's workspace (token buffers, patch matrix, side buffers) + its x_hat block
MIN_INFLIGHT = 2


def preflight_report(device, free_bytes=None):
    """What THIS rank found (no collective).  free_bytes: override for tests / CPU dry runs (None on a CPU device = not
    checked)."""
    rep = {"rank": int(os.environ.get("RANK", "0")), "n_cpus": len(os.sched_getaffinity(0)),
           "gpu_check": gpu_bind_check(), "free_bytes": free_bytes, "total_bytes": None, "device": str(device)}
    fake = os.environ.get("CRA5_TEST_FREE_GIB")          # tests: "r:gib,r:gib" - a rank whose GPU is mostly taken
    if fake:
        for part in fake.split(","):
            r, _, g = part.partition(":")
            if int(r) == rep["rank"]:
                rep["free_bytes"] = int(float(g) * GIB)
    fake = os.environ.get("CRA5_TEST_N_CPUS")
    if fake:
        for part in fake.split(","):
            r, _, c = part.partition(":")
            if int(r) == rep["rank"]:
                rep["n_cpus"] = int(c)
    if rep["free_bytes"] is None and torch.device(device).type == "cuda":
        free, total = torch.cuda.mem_get_info(device)
        rep["free_bytes"], rep["total_bytes"] = int(free), int(total)
    return rep


def plan_inflight(reports, requested, pool_frames):
    """Frames in flight for EVERY rank of the job (one figure: the ranks step in lock-step through barriers) from the
    gathered reports.  Pure function (tested on CPU).  Returns (inflight, [reasons])."""
    inflight, why = int(requested), []
    floor = min(MIN_INFLIGHT, inflight)        # (a caller that ASKS for one frame in flight - profiling passes - gets it)
    for r in reports:
        free = r.get("free_bytes")
        if free is not None:
            room = free - BASE_BYTES - pool_frames * FRAME_BYTES
            fit = int(room // PER_INFLIGHT_BYTES)
            if fit < inflight:
                why.append(f"rank {r['rank']}: {free / GIB:.1f} GiB of device memory free -> room for {max(fit, 0)} frames in "
                           f"flight beside the weights and a {pool_frames}-frame pool (requested {requested})")
                inflight = min(inflight, fit)
        ncpu = r.get("n_cpus")
        if ncpu is not None and ncpu < inflight:
            why.append(f"rank {r['rank']}: {ncpu} host CPUs in its mask < {inflight} frame threads (each frame's rANS phase "
                       "wants a core of its own)")
            inflight = max(floor, min(inflight, ncpu))     # (few cores slow the job down, they do not stop it)
    if inflight < floor:
        raise RuntimeError("preflight: the job cannot keep even %d frames in flight: %s" % (MIN_INFLIGHT, "; ".join(why)))
    return inflight, why


def preflight(device, requested, pool_frames, free_bytes=None):
    """Collective (every rank calls it, after init_from_env, before any model memory exists).  1. a 40-byte all-gather -
    the first RCCL traffic of the job, timed; 2. all_gather_object of the per-rank reports; 3. plan_inflight on the
    gathered list (identical on every rank).  Returns the dict bench.py puts into `config.preflight`."""
    import time
    t0 = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    probe = gather_stats([[rank, 1, 2, 3, 4]], torch.device(device))      # int64[world, 5]: 40 bytes per rank
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if probe[:, 0].tolist() != list(range(world)):
        raise RuntimeError(f"preflight: the 40-byte all-gather returned ranks {probe[:, 0].tolist()}, expected 0..{world - 1}")
    t1 = time.perf_counter()
    reports = sorted(gather_objects(preflight_report(device, free_bytes)), key=lambda r: r["rank"])
    inflight, why = plan_inflight(reports, requested, pool_frames)
    mism = [r["rank"] for r in reports if r["gpu_check"].get("match") is False]
    return {"first_collective_s": t1 - t0, "seconds": time.perf_counter() - t0, "inflight_requested": int(requested),
            "inflight": inflight, "lowered_because": why or None,
            "gpu_numa_assumption_mismatch_ranks": mism or None,
            "per_rank": [{"rank": r["rank"], "n_cpus": r["n_cpus"],
                          "free_gib": None if r["free_bytes"] is None else round(r["free_bytes"] / GIB, 1),
                          "gpu_check": r["gpu_check"]} for r in reports]}
