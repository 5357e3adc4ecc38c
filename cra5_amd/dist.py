"""Multi-GPU driver pieces: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The path shards by FRAME: hourly snapshots are independent units
(no cross-frame state, one rANS stream pair per frame - entropy_models.py:263-272 in the
reference), so weights are replicated and there is NO data-path collective.  The single
exchange is an all-gather of a tiny per-frame stats tensor (bytes of the y / z streams,
escape count, stream checksum): 32 B per frame, latency-bound, so ring-vs-tree or xGMI
link bandwidth is irrelevant for it.
"""
import os
import zlib

import torch
import torch.distributed as dist

STATS_FIELDS = ("frame", "y_bytes", "z_bytes", "crc32")


def init_from_env(device_type="cuda"):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)   # binds the RCCL communicator to this rank's GPU (eager init)
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


def gpu_numa_node(local):
    """NUMA node of GPU `local` from its PCI address (sysfs), or None when it cannot be told."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        return node if node >= 0 else None
    except Exception:  # noqa: BLE001
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


def bind_rank_to_numa(local, n_local):
    """Pin this process (all its future threads: the 12 frame threads, the rANS pool) to the CPUs of its
    GPU's NUMA node, shared equally with the other ranks whose GPU sits on the same node.  Returns a dict
    describing what was done (reported in bench.py's JSON line)."""
    info = {"numa_node": None, "cpus": None, "bound": False}
    try:
        nodes = [gpu_numa_node(i) for i in range(n_local)]
        node = nodes[local]
        node_cpus = None
        peers = None
        if node is not None:
            node_cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
            same = [i for i in range(n_local) if nodes[i] == node]
            peers = (same.index(local), len(same))
        cpus = rank_cpu_set(local, n_local, node, peers, None, node_cpus)
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(numa_node=node, cpus=len(cpus), bound=True, first_cpu=cpus[0], last_cpu=cpus[-1])
    except Exception as e:  # noqa: BLE001
        info["error"] = repr(e)
    return info


def shard_frames(n_frames, rank, world):
    """Contiguous block partition: rank r owns frames [lo, hi). 64 frames on 8 GPUs ->
    [8r, 8r+8).  Remainders go to the lowest ranks."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return range(lo, hi)


def frame_stats(frame_id, strings):
    """One int64 row per frame: (frame, y_bytes, z_bytes, crc32 of y||z)."""
    y, z = strings[0][0], strings[1][0]
    return [int(frame_id), len(y), len(z), zlib.crc32(z, zlib.crc32(y)) & 0xFFFFFFFF]


def gather_stats(rows, device):
    """all_gather of the per-rank stats (padded to the max per-rank frame count; -1 rows are
    padding).  Returns an int64 tensor [n_frames_total, 4] sorted by frame id on every rank."""
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
