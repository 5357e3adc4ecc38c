"""GPU busy / idle analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals over the
steady-state part of the run, gap statistics, idle time by following kernel.
   python tools/trace_gaps.py gpurun_out/prof_x/bench_kernel_trace.csv [start_fraction=0.5]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * frac
sel = [e for e in ev if e[0] >= lo]
busy, gaps = 0, []
cur_s, cur_e = sel[0][0], sel[0][1]
ksum = 0
for s, e, n in sel:
    ksum += e - s
for s, e, n in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = sel[-1][1] - sel[0][0]
print(f"window {wall/1e6:.1f} ms: GPU busy (union) {busy/1e6:.1f} ms = {100*busy/wall:.1f} %, sum of kernel durations {ksum/1e6:.1f} ms "
      f"(overlap factor {ksum/busy:.2f}), kernels {len(sel)}")
g = sorted(x[0] for x in gaps)
if g:
    print(f"gaps: n {len(g)}, median {statistics.median(g)/1e3:.1f} us, p90 {g[int(0.9*len(g))]/1e3:.1f} us, max {g[-1]/1e3:.1f} us, total {sum(g)/1e6:.1f} ms")
    by = collections.Counter()
    for d, n in gaps:
        by[n] += d
    print("idle time by the kernel that follows the gap (ms):")
    for n, d in by.most_common(8):
        print(f"  {d/1e6:8.2f}  {n}")
by_k = collections.Counter()
for s, e, n in sel:
    by_k[n] += e - s
print("kernel time (ms, sum of durations):")
for n, d in by_k.most_common(10):
    print(f"  {d/1e6:8.2f}  {n}")
