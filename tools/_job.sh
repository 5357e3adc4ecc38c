cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
CRA5_SHARE_GPU=1 CRA5_DIST_BACKEND=gloo python bench.py --gpus 4 --steps 3 --warmup 1 --settle-batches 0 --roofline-steps 1 --no-cpu-baseline --no-api-sample --no-f16-sample --no-kernel-timer --inflight 3 --frame-pool 2 > gpurun_out/r4g/share4.json 2> gpurun_out/r4g/share4.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4g/share4.json') if l.startswith('{')][-1])
for r in d['config']['host']['per_rank']: print(r)
PY
CRA5_LIB=build_variants/libcra5_grp4.so python tools/gemm_trace.py --active 2>&1 | grep -v Warn > gpurun_out/r4g/active_sweep.txt
grep -E "^act|shader clock|rate:" gpurun_out/r4g/active_sweep.txt
