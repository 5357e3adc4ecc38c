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


def test_bench_self_launches_two_ranks_dry():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls bench.py): it must re-launch
    itself as 2 ranks under torch.distributed.run, rendezvous, shard, all-gather, and rank 0 prints ONE JSON
    line; exit code 0.  --dry-dist = the same code path up to the GPU work, on gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--dry-dist"], cwd=root, env=env, capture_output=True, text=True, timeout=600,
                       stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["self_launched"] is True and d["backend"] == "gloo"
    assert d["stats_rows"] == 10 and d["frames_of_rank0"] == [0, 5]


def test_bench_refuses_world_size_mismatch():
    """Under a launcher whose WORLD_SIZE disagrees with --gpus the job must stop, not measure something else."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--dry-dist"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_rank_cpu_sets_partition_a_numa_node():
    """8 ranks on a 2-socket EPYC (node0 = 0-63,128-191; node1 = 64-127,192-255), GPUs 0-3 on node 0: every
    rank gets 16 whole cores (core + SMT sibling), disjoint from every other rank's."""
    nodes = {0: D._parse_cpulist("0-63,128-191"), 1: D._parse_cpulist("64-127,192-255")}
    seen = set()
    for local in range(8):
        node = local // 4
        cpus = D.rank_cpu_set(local, 8, node, (local % 4, 4), range(256), nodes[node])
        assert len(cpus) == 32 and not (seen & set(cpus))
        assert all((c + 128) in cpus for c in cpus if c < 128)     # whole cores
        assert set(cpus) <= set(nodes[node])
        seen |= set(cpus)
    assert len(seen) == 256
    # unknown topology: equal contiguous shares of what the process may run on
    assert D.rank_cpu_set(1, 2, None, None, range(8), None) == [4, 5, 6, 7]
    assert D._parse_cpulist("0-2,5,7-8\n") == [0, 1, 2, 5, 7, 8]
