cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench20_a.json 2> gpurun_out/r4a/bench20_a.err
python tools/power_probe.py 4 > gpurun_out/r4a/power.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4a/bench20_b.json 2>> gpurun_out/r4a/bench20_a.err
python bench.py --no-cpu-baseline --no-api-sample > gpurun_out/r4a/bench96.json 2>> gpurun_out/r4a/bench20_a.err
tail -3 gpurun_out/r4a/power.txt
