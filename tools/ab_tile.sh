cd $GRAFT_REPO_ROOT
CRA5_GEMM_TILE=193 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm 2>&1 | tail -2
for rep in 1 2; do
for t in 0 193; do
  echo "tile $t"; CRA5_GEMM_TILE=$t timeout 200 python tools/gemm_bench.py 2>&1 | tail -1
done; done
