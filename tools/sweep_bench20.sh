run() { echo -n "$1 :: "; shift; env "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2), d['warmup_settle_frames'])"; }
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer"
run "default" $B
run "hwq4 kernarg0" GPU_MAX_HW_QUEUES=4 HIP_FORCE_DEV_KERNARG=0 $B
run "settle0" $B --settle-batches 0
run "settle0 again" $B --settle-batches 0
run "steps40" python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timer
run "pool2" $B --frame-pool 2
