cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or unembed" 2>&1 | tail -3
for rep in 1 2; do
  timeout 300 python tools/gemm_bench.py 2>&1 | tail -1
  CRA5_LIB=build_variants/libcra5_ghead.so timeout 300 python tools/gemm_bench.py 2>&1 | tail -1
done
for rep in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case > gpurun_out/r4l/new_$rep.json 2>> gpurun_out/r4l/err.txt
python tools/bench_line.py gpurun_out/r4l/new_$rep.json
CRA5_LIB=build_variants/libcra5_ghead.so python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case > gpurun_out/r4l/head_$rep.json 2>> gpurun_out/r4l/err.txt
python tools/bench_line.py gpurun_out/r4l/head_$rep.json
done
