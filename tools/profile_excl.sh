#!/bin/bash
# GPU box: rocprofv3 kernel stats of bench.py with exclusive GPU phases (every launch alone on the chip).
#   tools/profile_excl.sh TAG [extra bench args]   -> gpurun_out/TAG/{bench_kernel_stats.csv, bench.log}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf $R/gpurun_out/$tag; mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag/prof -o bench -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-api-sample --exclusive "$@" > $R/gpurun_out/$tag/bench.log 2>&1 < /dev/null
cd $R
f=$(find gpurun_out/$tag/prof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/$tag/bench_kernel_stats.csv
rm -rf gpurun_out/$tag/prof
grep -h '"metric"' gpurun_out/$tag/bench.log | cut -c1-200
python tools/kstats.py gpurun_out/$tag/bench_kernel_stats.csv 30
