#!/bin/bash
# N consecutive runs of the driver's bench command on one box: frames/s, settle frames, wall seconds
for i in $(seq 1 ${1:-3}); do
  t0=$(date +%s.%N)
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'fps; settle frames', d['warmup_settle_frames'], '; roofline', round(d['roofline']['frac'],3))"
  echo "  wall $(echo "$(date +%s.%N) - $t0" | bc) s"
done
