#!/bin/bash
# GPU box: rocprofv3 kernel stats of the bench command + HBM traffic PMC passes (separate runs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_excl -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --exclusive --inflight 2 > $R/gpurun_out/prof_excl_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --exclusive --inflight 1 --no-kernel-timer > $R/gpurun_out/pmc_traffic_$c.log 2>&1
done
grep -h '"metric"' $R/gpurun_out/prof_final_bench.log $R/gpurun_out/prof_excl_bench.log | cut -c1-400
