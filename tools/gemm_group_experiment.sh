#!/bin/bash
# GPU box: tile-order experiment of tools/build_gemm_group_variants.sh - per variant: launch times (gemm_bench), the
# in-kernel shader clock + per-block phases (gemm_trace), FETCH_SIZE per launch (rocprofv3 --pmc, its own pass).
R=$GRAFT_REPO_ROOT
cd $R
out=gpurun_out/grp_exp; mkdir -p $out
for rep in 1 2; do
  for G in "$@"; do
    echo -n "G=$G " ; CRA5_LIB=build_variants/libcra5_grp$G.so timeout 300 python tools/gemm_bench.py 2>&1 | tail -1
  done
done > $out/bench.txt 2>&1
for G in "$@"; do
  echo "=== G=$G"; CRA5_LIB=build_variants/libcra5_grp$G.so timeout 300 python tools/gemm_trace.py 2>&1 | grep -v Warning
done > $out/trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for G in "$@"; do
  CRA5_LIB=$R/build_variants/libcra5_grp$G.so timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_$G -o pmc -- python $R/tools/gemm_bench.py --once > $R/$out/pmc_$G.log 2>&1
done
cd $R
python - "$@" <<'PY' > $out/fetch.txt 2>&1
import csv, glob, sys, collections
for G in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/grp_exp/pmc_fetch_{G}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_nt_split_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
                agg[r['Grid_Size']].append(float(r['Counter_Value']))
    # FETCH_SIZE: KB, x2 on gfx950 (MI355X_MICROARCH.md)
    print('G=%s' % G, ' '.join('grid %s: %.0f MB (n=%d)' % (g, 2 * sum(v) / len(v) * 1024 / 1e6, len(v)) for g, v in sorted(agg.items(), key=lambda kv: int(kv[0]))))
PY
cat $out/bench.txt $out/fetch.txt; grep -E "===|shader clock|^qkv|^fc1|^fc2|^proj|per block" $out/trace.txt
