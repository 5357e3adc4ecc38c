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
        kw = {}
        if device_type == "cuda":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)   # binds the RCCL communicator to this rank's GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


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
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)
