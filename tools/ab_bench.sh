# usage: tools/ab_bench.sh VARIANT [VARIANT ...]   (GPU box) interleaved tools/gemm_bench.py: default library, then each variant
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  timeout 200 python tools/gemm_bench.py 2>&1 | tail -1
  for v in "$@"; do CRA5_LIB=build_variants/libcra5_$v.so timeout 200 python tools/gemm_bench.py 2>&1 | tail -1; done
done
