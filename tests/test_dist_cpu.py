"""N > 1 path on CPU: world_size-2 gloo processes shard frames and all-gather the
per-frame bitstream stats exactly as bench.py does over RCCL."""
import os
import socket

import torch
import torch.multiprocessing as mp

from cra5_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("cpu")
    rows = []
    for f in D.shard_frames(n_frames, r, w):
        y = bytes([f % 251]) * (100 + f)       # stand-in streams: the stats only look at bytes
        z = bytes([(f * 7) % 251]) * (10 + f)
        rows.append(D.frame_stats(f, [[y], [z]]))
    D.barrier()
    stats = D.gather_stats(rows, torch.device("cpu"))
    t = D.max_over_ranks(1.0 + r, torch.device("cpu"))
    q.put((r, stats.tolist(), t))
    torch.distributed.destroy_process_group()


def test_shard_frames_partition():
    for n, w in ((64, 8), (10, 4), (3, 8), (0, 2)):
        parts = [list(D.shard_frames(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
    assert list(D.shard_frames(64, 3, 8)) == list(range(24, 32))  # rank r owns [8r, 8r+8)


def test_two_rank_gloo_allgather():
    world, n_frames = 2, 7  # uneven split: 4 + 3 (exercises the padding rows)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = []
    import zlib
    for f in range(n_frames):
        y, z = bytes([f % 251]) * (100 + f), bytes([(f * 7) % 251]) * (10 + f)
        expect.append([f, len(y), len(z), zlib.crc32(z, zlib.crc32(y)) & 0xFFFFFFFF])
    for r, stats, t in res:
        assert stats == expect      # every rank sees every frame, sorted
        assert t == 2.0             # max over ranks


def test_single_process_passthrough():
    rows = [D.frame_stats(1, [[b"ab"], [b"c"]]), D.frame_stats(0, [[b"x"], [b""]])]
    s = D.gather_stats(rows, torch.device("cpu"))
    assert s[:, 0].tolist() == [0, 1] and s[1, 1:3].tolist() == [2, 1]
