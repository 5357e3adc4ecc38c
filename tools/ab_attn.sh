# usage: tools/ab_attn.sh VARIANT ...   (GPU box) attention tests with the default library, then interleaved tools/attn_bench.py
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2
for rep in 1 2; do
  timeout 200 python tools/attn_bench.py 2>&1 | grep "split" | tr '\n' ' '; echo
  for v in "$@"; do echo -n "$v: "; CRA5_LIB=build_variants/libcra5_$v.so timeout 200 python tools/attn_bench.py 2>&1 | grep "split" | tr '\n' ' '; echo; done
done
