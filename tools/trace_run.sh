cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r3h; mkdir -p $R/gpurun_out/r3h
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3h/prof -o bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-api-sample --no-kernel-timer > $R/gpurun_out/r3h/bench.log 2>&1 < /dev/null
cd $R
f=$(find gpurun_out/r3h/prof -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f 0.5 | head -8
python tools/trace_concurrency.py $f > gpurun_out/r3h/concurrency.txt 2>&1; cat gpurun_out/r3h/concurrency.txt
rm -rf gpurun_out/r3h/prof
grep -h '"metric"' gpurun_out/r3h/bench.log | cut -c1-160
