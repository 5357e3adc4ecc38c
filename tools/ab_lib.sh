# usage: tools/ab_lib.sh VARIANT [VARIANT ...]   (GPU box) gemm tests under each variant, then interleaved gemm_bench
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== tests $v"; CRA5_LIB=build_variants/libcra5_$v.so timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm 2>&1 | tail -2
done
for rep in 1 2; do
  timeout 200 python tools/gemm_bench.py 2>&1 | tail -1
  for v in "$@"; do CRA5_LIB=build_variants/libcra5_$v.so timeout 200 python tools/gemm_bench.py 2>&1 | tail -1; done
done
