#!/bin/bash
# GPU box: the reduced-precision mode (BASELINE configs[4], CRA5_PRECISION=f16) under rocprofv3 - kernel stats of the
# exclusive bench command, the MFMA-pipe PMC pass and the two HBM-traffic PMC passes.   tools/profile_f16.sh [ROUND=r05] -> gpurun_out/ROUND/ROUND_f16_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r05}
O=$R/gpurun_out/$T
mkdir -p $O
COMMON="--precision f16 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-matched-sample"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16 -o bench -- python $R/bench.py --steps 16 --warmup 4 --exclusive $COMMON > $O/bench_f16_excl.log 2>&1 < /dev/null
cp $(find $O/prof_f16 -name '*kernel_stats.csv' | head -1) $O/${T}_f16_bench_exclusive_kernel_stats.csv
grep -h '"metric"' $O/bench_f16_excl.log | tail -1 > $O/${T}_f16_bench_exclusive.json
rm -rf $O/prof_f16
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f16 -o pmc -- python $R/bench.py --steps 1 --warmup 1 --exclusive --inflight 1 --no-kernel-timer $COMMON > $O/pmc_f16.log 2>&1 < /dev/null
cd $R
python - "$O" "$T" <<'PY'
import collections, csv, json, sys
O, T = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{O}/pmc_f16/pmc_counter_collection.csv")):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "tools/profile_f16.sh: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE of `python bench.py "
                 "--precision f16 --steps 1 --warmup 1 --exclusive --inflight 1 --no-kernel-timer`",
       "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs)", "per_kernel": {}}
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    e = {"launches": len(next(iter(c.values()))), "GRBM_GUI_ACTIVE": gui,
         "mfma_util": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)}
    rows.append((e["launches"] * gui, k, e))
for _, k, e in sorted(rows, reverse=True)[:12]:
    out["per_kernel"][k] = e
    print(f"{k[:66]:66s} n {e['launches']:4d} mfma_util {e['mfma_util']:.3f}")
json.dump(out, open(f"{O}/{T}_f16_mfma_util.json", "w"), indent=1)
PY
rm -rf $O/pmc_f16
# HBM-side bytes per kernel in this mode (one counter per pass, --kernel-trace only) -> ${T}_f16_traffic.json
F=$R/gpurun_out/${T}_f16tr; rm -rf $F; mkdir -p $F
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $F/pmc_traffic_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --exclusive --inflight 1 --no-kernel-timer $COMMON > $F/pmc_traffic_$c.log 2>&1 < /dev/null
done
cd $R
python tools/traffic_summary.py $F $O/${T}_f16 "--precision f16" | tail -8
rm -rf $F
cut -c1-160 $O/${T}_f16_bench_exclusive.json
