cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout 1800 python -m pytest tests/test_model_gpu.py -q -x -k "compact" 2>&1 | tail -40 > gpurun_out/r4j/tests.txt
cat gpurun_out/r4j/tests.txt | cut -c1-250
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-sample --no-f16-sample --inflight 8 > gpurun_out/r4j/b8.json 2> gpurun_out/r4j/err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4j/b8.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['achieved'], d['roofline']['launches'], d['roofline']['avg_launch_ms'], d['roofline']['gemm_ms_per_step'])
print(d.get('roofline_timed_region'))
print(d.get('attention'))
PY
