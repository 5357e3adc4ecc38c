# GPU box: kernel trace of the overlapping pipeline -> concurrency summary (tools/trace_concurrency.py) in gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r05t}
X=${2:-}   # extra bench flags, e.g. "--precision f16"
rm -rf $R/gpurun_out/$T; mkdir -p $R/gpurun_out/$T
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T/prof -o bench -- python $R/bench.py $X --steps 40 --warmup 5 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-kernel-timer --no-matched-sample --no-clock-sampler > $R/gpurun_out/$T/bench.log 2>&1 < /dev/null
cd $R
f=$(find gpurun_out/$T/prof -name '*kernel_trace.csv' | head -1)
python tools/trace_gaps.py $f 0.5 | head -8 > gpurun_out/$T/concurrency.txt
python tools/trace_concurrency.py $f >> gpurun_out/$T/concurrency.txt 2>&1; cat gpurun_out/$T/concurrency.txt
rm -rf gpurun_out/$T/prof
grep -h '"metric"' gpurun_out/$T/bench.log | cut -c1-160
