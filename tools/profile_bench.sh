#!/bin/bash
# GPU box: rocprofv3 kernel stats of the bench command + HBM traffic PMC passes (separate runs).
#   tools/profile_bench.sh [ROUND=r03]  -> gpurun_out/ROUND/ : copy the *_kernel_stats.csv / *.json summaries into profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r05}
O=$R/gpurun_out/$T
rm -rf $O; mkdir -p $O
# 1. the default bench command (overlapping GPU phases, 12 frames in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o bench -- python $R/bench.py --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-matched-sample > $O/bench_default.log 2>&1 < /dev/null
# 2. exclusive GPU phases: every launch alone on the chip (the durations the roofline object uses)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_excl -o bench -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-matched-sample --exclusive > $O/bench_excl.log 2>&1 < /dev/null
# 3. HBM-side bytes per kernel: one counter per pass, --kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_traffic_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-matched-sample --exclusive --inflight 1 --no-kernel-timer > $O/pmc_traffic_$c.log 2>&1 < /dev/null
done
cd $R
cp $(find $O/prof_default -name '*kernel_stats.csv' | head -1) $O/${T}_bench_default_kernel_stats.csv
cp $(find $O/prof_excl -name '*kernel_stats.csv' | head -1) $O/${T}_bench_exclusive_kernel_stats.csv
grep -h '"metric"' $O/bench_default.log | tail -1 > $O/${T}_bench_default.json
grep -h '"metric"' $O/bench_excl.log | tail -1 > $O/${T}_bench_exclusive.json
python tools/trace_gaps.py $(find $O/prof_default -name '*kernel_trace.csv' | head -1) 0.5 | head -4
python tools/traffic_summary.py $O $O/$T
rm -rf $O/prof_default $O/prof_excl
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $O/pmc_traffic_$c; done
cut -c1-200 $O/${T}_bench_default.json; cut -c1-200 $O/${T}_bench_exclusive.json
python - "$O/${T}_bench_exclusive_kernel_stats.csv" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:16]
for r in rows:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms  {int(float(r['Calls'])):6d} x {float(r['TotalDurationNs']) / float(r['Calls']) / 1e3:8.1f} us  {r['Name'][:100]}")
PY
