#!/bin/bash
# usage: tools/bench_sweep.sh REPEATS "args A" "args B" ...   (GPU box) - median frames/s per configuration
rep=$1; shift
for a in "$@"; do
  vals=""
  for i in $(seq $rep); do
    timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer $a > gpurun_out/b_sw.json 2>gpurun_out/b_sw.err < /dev/null
    v=$(python -c "import json; print(round(json.loads(open('gpurun_out/b_sw.json').read().strip().split('\n')[-1])['value'],2))")
    vals="$vals $v"
  done
  python -c "import sys,statistics; v=[float(x) for x in sys.argv[2:]]; print(f'{sys.argv[1]:50s} median {statistics.median(v):6.2f}  all {v}')" "$a" $vals
done
