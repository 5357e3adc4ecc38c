cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm 2>&1 | tail -2
for rep in 1 2; do
for v in "" build_variants/libcra5_g_group0.so build_variants/libcra5_g_group8.so; do
  CRA5_LIB=$v timeout 200 python tools/gemm_bench.py 2>&1 | tail -1
done; done
cd /tmp && export TMPDIR=/tmp
for v in default g_group0 g_group8; do
  lib=""; [ $v != default ] && lib=$GRAFT_REPO_ROOT/build_variants/libcra5_$v.so
  CRA5_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_group_$v -o pmc -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --once > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for v in ("default","g_group0","g_group8"):
    f = glob.glob(f"gpurun_out/pmc_group_{v}/**/*counter_collection.csv", recursive=True)
    if not f: print(v, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm_nt_split" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
    # group by dispatch order
    vals = [float(r["Counter_Value"]) for r in rows]
    print(v, len(vals), [round(x/1e3*64/1e3) for x in vals])
PY
