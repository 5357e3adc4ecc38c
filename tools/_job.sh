cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4k/gpu_tests.txt
tail -4 gpurun_out/r4k/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_bench.sh r04 > gpurun_out/r04_profile.log 2>&1
bash tools/pmc_mfma.sh r04 > gpurun_out/r04_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/r04_bench_unprofiled.json 2> gpurun_out/r04/unprof.err
python tools/bench_line.py gpurun_out/r04/r04_bench_unprofiled.json
tail -19 gpurun_out/r04_profile.log | cut -c1-200; tail -8 gpurun_out/r04_mfma.log
