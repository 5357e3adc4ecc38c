cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r4b/model_tests.txt
tail -30 gpurun_out/r4b/model_tests.txt
