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
        rows.append(D.frame_stats(f, [[y], [z]], n_escape=3 * f))
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
        expect.append([f, len(y), len(z), zlib.crc32(z, zlib.crc32(y)) & 0xFFFFFFFF, 3 * f])
    for r, stats, t in res:
        assert stats == expect      # every rank sees every frame, sorted
        assert t == 2.0             # max over ranks


def test_single_process_passthrough():
    rows = [D.frame_stats(1, [[b"ab"], [b"c"]]), D.frame_stats(0, [[b"x"], [b""]])]
    s = D.gather_stats(rows, torch.device("cpu"))
    assert s[:, 0].tolist() == [0, 1] and s[1, 1:3].tolist() == [2, 1]
    assert s.shape[1] == len(D.STATS_FIELDS) == 5 and D.STATS_FIELDS[4] == "n_escape"
    assert s[:, 4].tolist() == [-1, -1]                                                   # not counted by this caller


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
    assert d["stats_fields"] == list(D.STATS_FIELDS)


def test_bench_self_launches_eight_ranks_dry():
    """The driver's `python bench.py --gpus 8` shape: 8 ranks self-launched, each pinned to its
    own share of the host cores BEFORE its process group exists, 8 x K frames sharded and gathered, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "1",
                        "--dry-dist"], cwd=root, env=env, capture_output=True, text=True, timeout=900,
                       stdin=subprocess.DEVNULL)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["stats_rows"] == 64 and d["frames_of_rank0"] == [0, 8]     # BASELINE configs[3]: 8 per rank
    hosts = d["host_per_rank"]
    assert sorted(h["rank"] for h in hosts) == list(range(8))
    n_allowed = len(os.sched_getaffinity(0))
    if n_allowed >= 8:
        # every rank bound, to a set disjoint from every other rank's, and none of its threads outside its mask
        assert d["host_cpu_sets_disjoint"] is True
        assert all(h["numa_bind"] and h["numa_bind"]["bound"] for h in hosts)
        assert sum(h["n_cpus"] for h in hosts) <= n_allowed
    assert all(h["threads_outside_mask"] == 0 for h in hosts)


def test_plan_rank_cpus_one_rule_for_all_ranks():
    """when the NUMA node of ANY rank's GPU is unknown every rank falls back to the same rule (equal shares of
    the allowed CPUs) - a node-bound rank and a fallback rank must never overlap."""
    nodes_cpus = {0: D._parse_cpulist("0-63,128-191"), 1: D._parse_cpulist("64-127,192-255")}
    known = [0, 0, 0, 0, 1, 1, 1, 1]
    mixed = [0, 0, None, 0, 1, 1, 1, 1]
    for nodes in (known, mixed):
        seen = set()
        for local in range(8):
            node, cpus = D.plan_rank_cpus(local, 8, nodes, range(256), lambda n: nodes_cpus[n])
            assert cpus and not (seen & set(cpus))
            seen |= set(cpus)
            assert (node is None) == (None in nodes)
        assert len(seen) == 256


def test_gpu_bdf_lookup_without_hip(tmp_path, monkeypatch):
    """The GPU's PCI address comes from the KFD topology in sysfs (no HIP call: the bind runs before the runtime
    starts) and honours HIP_VISIBLE_DEVICES remapping."""
    root = tmp_path / "nodes"
    for i, (simd, loc) in enumerate(((0, 0), (0, 0), (1024, 0x0500), (1024, 0x1508), (1024, 0x6500))):
        (root / str(i)).mkdir(parents=True)
        (root / str(i) / "properties").write_text(f"cpu_cores_count 64\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    bdfs = D.kfd_gpu_bdfs(str(root))
    assert bdfs == ["0000:05:00.0", "0000:15:01.0", "0000:65:00.0"]
    pci = tmp_path / "pci"
    for b, n in zip(bdfs, (0, 0, 1)):
        (pci / b).mkdir(parents=True)
        (pci / b / "numa_node").write_text(f"{n}\n")
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert [D.gpu_numa_node(i, bdfs, str(pci)) for i in range(3)] == [0, 0, 1]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert [D.gpu_numa_node(i, bdfs, str(pci)) for i in range(2)] == [1, 0]
    assert D.gpu_numa_node(2, bdfs, str(pci)) is None          # not visible


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


def test_visible_gpu_count(monkeypatch):
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert D.visible_gpu_count(8) == 8 and D.visible_gpu_count(0) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0,5")
    assert D.visible_gpu_count(8) == 3
    assert D.visible_gpu_count(4) == 2          # an entry that names no GPU ends the list, like the runtime's own rule


def test_bench_committed_reference_reads_the_profile_of_its_own_precision():
    """`roofline.committed_reference` comes from the newest committed rocprofv3 summary of the SAME arithmetic: the
    fp32-accurate line must not pick up `rNN_f16_bench_exclusive_kernel_stats.csv` (it sorts after
    `rNN_bench_exclusive_...` and holds half-length launches), and the reduced-precision line is priced against the
    plain-f16 peak."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, root)
    import bench
    a, b = bench.rocprof_gemm_frac(), bench.rocprof_gemm_frac(f16=True)
    assert a is not None and "_f16_" not in a["file"] and 0.2 < a["frac"] < 1.0
    assert b is None or ("_f16_" in b["file"] and 0.05 < b["frac"] < 1.0)
    if b is not None:
        assert b["avg_launch_ms"] < a["avg_launch_ms"]


# ---- round 6: first-contact hardening of the N-rank job -------------------------------------------------


def _dry(n, extra_env=None, steps=4, timeout=900):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                            "MASTER_PORT", "LOCAL_WORLD_SIZE", "CRA5_JOB_DIR")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", "1",
                        "--dry-dist"], cwd=root, env=env, capture_output=True, text=True, timeout=timeout,
                       stdin=subprocess.DEVNULL)
    return p, [l for l in p.stdout.split("\n") if l.startswith("{")]


def test_plan_inflight_pure():
    """One frames-in-flight figure for the whole job from the gathered per-rank reports."""
    ok = [dict(rank=r, free_bytes=int(280 * D.GIB), n_cpus=32) for r in range(8)]
    assert D.plan_inflight(ok, 12, 8) == (12, [])
    tight = [dict(r) for r in ok]
    tight[3]["free_bytes"] = int(30 * D.GIB)          # somebody else's process sits on rank 3's GPU
    n, why = D.plan_inflight(tight, 12, 8)
    room = 30 * D.GIB - D.BASE_BYTES - 8 * D.FRAME_BYTES
    assert n == int(room // D.PER_INFLIGHT_BYTES) and 2 <= n < 12 and len(why) == 1 and why[0].startswith("rank 3")
    few = [dict(r) for r in ok]
    few[6]["n_cpus"] = 5                              # a cgroup that leaves rank 6 five CPUs
    n, why = D.plan_inflight(few, 12, 8)
    assert n == 5 and "rank 6" in why[0]
    assert D.plan_inflight(ok, 1, 1) == (1, [])       # one frame in flight on request (profiling passes): not a failure
    none = [dict(rank=0, free_bytes=None, n_cpus=None)]          # CPU dry run: nothing to check
    assert D.plan_inflight(none, 12, 8) == (12, [])
    import pytest
    dead = [dict(rank=0, free_bytes=int(4 * D.GIB), n_cpus=64)]
    with pytest.raises(RuntimeError, match="cannot keep even"):
        D.plan_inflight(dead, 12, 24)
    # the 12-in-flight pipeline of the 1-GPU bench (24-frame pool) fits the soak's 61.8 GiB figure with room to spare
    need = D.BASE_BYTES + 24 * D.FRAME_BYTES + 12 * D.PER_INFLIGHT_BYTES
    assert 55 * D.GIB < need < 75 * D.GIB


def test_eight_rank_preflight_lowers_inflight_and_reports_per_rank():
    """8 gloo ranks: the preflight's 40-byte all-gather, the gathered reports, ONE frames-in-flight figure for all ranks
    when one rank's GPU has little memory free, and a `per_rank` row from every rank."""
    import json
    p, lines = _dry(8, {"CRA5_TEST_FREE_GIB": "3:30,5:250", "CRA5_INFLIGHT": "12",
                        "CRA5_TEST_N_CPUS": ",".join(f"{r}:32" for r in range(8))}, steps=8)   # (as on a 2 x 64-core node)
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    pre = d["preflight"]
    assert pre["inflight_requested"] == 12 and 2 <= pre["inflight"] < 12
    assert pre["lowered_because"] and pre["lowered_because"][0].startswith("rank 3")
    assert [r["rank"] for r in pre["per_rank"]] == list(range(8))
    assert pre["per_rank"][3]["free_gib"] == 30.0 and pre["per_rank"][5]["free_gib"] == 250.0
    assert pre["first_collective_s"] >= 0
    assert sorted(r["rank"] for r in d["per_rank"]) == list(range(8))
    assert all(r["value"] and r["value"] > 0 and "host_phase_ms" in r and "shader_clock" in r for r in d["per_rank"])


def test_eight_rank_job_with_a_dying_rank_prints_one_error_line():
    """A rank that dies before the timed region: the job's stdout is ONE parsable JSON line with "error", the failing
    rank and its traceback; the exit code is non-zero (the driver records the failure instead of a missing line)."""
    import json
    p, lines = _dry(8, {"CRA5_TEST_FAIL_RANK": "5"}, steps=8)
    assert p.returncode != 0
    assert len(lines) == 1, p.stdout + p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 8 and d["failed_rank"] == 5 and d["stage"] == "warm-up"
    assert "injected failure on rank 5" in d["error"] and "RuntimeError" in d["traceback"] and "_test_failure_hook" in d["traceback"]


def test_single_rank_refusal_is_an_error_line_too():
    """`--gpus 2` under a launcher that says WORLD_SIZE=1: refused (as before) AND stdout carries the error line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    env.pop("CRA5_JOB_DIR", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--dry-dist"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert p.returncode != 0
    lines = [l for l in p.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1 and "WORLD_SIZE=1" in json.loads(lines[0])["error"]
