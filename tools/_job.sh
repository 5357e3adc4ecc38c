cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r04 > gpurun_out/r04_profile.log 2>&1
bash tools/pmc_mfma.sh r04 > gpurun_out/r04_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/r04_bench_unprofiled.json 2> gpurun_out/r04/unprof.err
python tools/bench_line.py gpurun_out/r04/r04_bench_unprofiled.json
timeout 300 python -m pytest tests/test_bench_gpu.py -q -k contract 2>&1 | tail -2
tail -19 gpurun_out/r04_profile.log | cut -c1-200; tail -8 gpurun_out/r04_mfma.log
