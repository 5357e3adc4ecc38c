#!/bin/bash
# GPU box: rocprofv3 kernel stats of the bench command + HBM traffic PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1. the default bench command (shared streams, 8 frames in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_default -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_r02_default_bench.log 2>&1 < /dev/null
# 2. exclusive GPU phases: every launch alone on the chip (the durations the roofline object uses)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_excl -o bench -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --exclusive > $R/gpurun_out/prof_r02_excl_bench.log 2>&1 < /dev/null
# 3. HBM-side bytes per kernel: one counter per pass, --kernel-trace only
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --exclusive --inflight 1 --no-kernel-timer > $R/gpurun_out/pmc_traffic_$c.log 2>&1 < /dev/null
done
cd $R
grep -h '"metric"' gpurun_out/prof_r02_default_bench.log gpurun_out/prof_r02_excl_bench.log | cut -c1-300
python tools/trace_gaps.py gpurun_out/prof_r02_default/bench_kernel_trace.csv 0.5 | head -4
python tools/traffic_summary.py gpurun_out gpurun_out/r02
