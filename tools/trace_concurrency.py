"""How many kernels run at the same time, and which ones run ALONE, from a rocprofv3 --kernel-trace CSV (second half of
the run): time at concurrency level 0 / 1 / 2 / 3+, and for the big kernels the share of their duration spent alone on
the chip (nothing to fill a partial round / a tail with).
    python tools/trace_concurrency.py bench_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
marks = [e for e in ev if "clock_stamp_kernel" in e[2]]
if len(marks) >= 2:
    # bench.py brackets its TIMED REGION with two clock_stamp_kernel launches: everything below is about that region only
    lo, hi = marks[0][1], marks[1][0]
    sel = [e for e in ev if e[0] >= lo and e[1] <= hi and "clock_stamp_kernel" not in e[2] and "clock_probe_kernel" not in e[2]]
    print(f"timed region (between the two clock_stamp_kernel marks): {(hi - lo) / 1e6:.1f} ms, {len(sel)} kernels")
else:
    lo = t0 + (t1 - t0) * 0.5
    sel = [e for e in ev if e[0] >= lo]
pts = []
for i, (s, e, n) in enumerate(sel):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
level_time = collections.Counter()
alone = collections.Counter()
total = collections.Counter()
active = set()
prev = pts[0][0]
for t, d, i in pts:
    dt = t - prev
    if dt > 0:
        level_time[min(len(active), 4)] += dt
        for j in active:
            total[j] += dt
            if len(active) == 1:
                alone[j] += dt
    prev = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
wall = pts[-1][0] - pts[0][0]
if len(marks) >= 2:
    # idle = no kernel running, measured against the whole marked region (incl. its edges)
    busy = sum(v for k, v in level_time.items() if k > 0)
    gaps = []
    cur_end = lo
    for s_, e_, n_ in sel:
        if s_ > cur_end:
            gaps.append((s_ - cur_end, n_))
        cur_end = max(cur_end, e_)
    if hi > cur_end:
        gaps.append((hi - cur_end, "<end of region>"))
    gaps.sort(reverse=True)
    print(f"GPU busy (union of kernels) {busy / 1e6:.1f} ms = {100 * busy / (hi - lo):.1f} % of the region; idle {100 - 100 * busy / (hi - lo):.1f} %; "
          f"largest gaps (us, kernel that follows): " + ", ".join(f"{g / 1e3:.0f} {n[:40]}" for g, n in gaps[:5]))
print("concurrency level -> share of the wall clock:", {k: round(100 * v / wall, 1) for k, v in sorted(level_time.items())})


def short(n):
    for key in ("gemm_nt_split_kernel<2, 4, 4", "gemm_nt_split_kernel<2, 4, 3", "window_attention_split_kernel<12", "window_attention_split_kernel<4, false, false",
                "window_attention_split_kernel<4, false, true", "layernorm", "im2col_tiled", "col2im_tiled", "copyBuffer", "small_gemm", "hyper_attention"):
        if key in n:
            return key
    return "other"


by_a, by_t, cnt = collections.Counter(), collections.Counter(), collections.Counter()
for i, (s, e, n) in enumerate(sel):
    k = short(n)
    by_a[k] += alone[i]
    by_t[k] += e - s
    cnt[k] += 1
print(f"{'kernel':48s} {'launches':>8s} {'avg us':>9s} {'alone %':>8s} {'sum ms':>8s}")
for k, v in by_t.most_common():
    print(f"{k:48s} {cnt[k]:8d} {v / cnt[k] / 1e3:9.1f} {100 * by_a[k] / max(v, 1):8.1f} {v / 1e6:8.1f}")
