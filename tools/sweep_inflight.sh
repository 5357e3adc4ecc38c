# (GPU box) frames in flight x GPU-phase slots on the driver's 20-step region, each configuration twice
for rep in 1 2; do
for cfg in "8 3" "10 3" "12 3" "12 2" "12 4" "14 3" "16 3"; do set -- $cfg
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --inflight $1 --gpu-slots $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 slots $2', round(d['value'],2), round(d['ms_per_step'],2))"
done; done
