cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
CRA5_SHARE_GPU=1 CRA5_DIST_BACKEND=gloo python bench.py --gpus 4 --steps 3 --warmup 1 --settle-batches 0 --roofline-steps 1 --no-cpu-baseline --no-api-sample --no-f16-sample --no-kernel-timer --inflight 3 --frame-pool 2 > gpurun_out/r4d/share4.json 2> gpurun_out/r4d/share4.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4d/share4.json') if l.startswith('{')][-1])
print(json.dumps(d['config']['host'], indent=0)[:3000])
PY
bash tools/gemm_group_experiment.sh 1 2 4 8 16 64
