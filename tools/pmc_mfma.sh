#!/bin/bash
# GPU box: MFMA-pipe utilisation per kernel (separate PMC pass, --kernel-trace only) -> gpurun_out/ROUND/ROUND_mfma_util.json
#   tools/pmc_mfma.sh [ROUND=r03]
T=${1:-r05}; export CRA5_PROF_TAG=$T; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-api-sample --no-f16-sample --no-best-case --no-matched-sample --exclusive --inflight 1 --no-kernel-timer > $R/gpurun_out/pmc_mfma.log 2>&1 < /dev/null
cd $R
python - <<'PY'
import collections, csv, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_mfma/pmc_counter_collection.csv")):
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU "
                 "SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE of `python bench.py --steps 1 --warmup 1 --exclusive --inflight 1 "
                 "--no-kernel-timer` (tools/pmc_mfma.sh)",
       "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs).  Checked against known work: "
                  "SQ_VALU_MFMA_BUSY_CYCLES is the whole-chip sum and equals 32 cycles x the MFMA count exactly (qkv launch: 1.935e8 = "
                  "3 * 10496 * 3072 * 1024 / 16384 * 32), GRBM_GUI_ACTIVE is summed over the 8 XCDs (3.1e6 for a ~190 us launch).  "
                  "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles",
       "per_kernel": {}}
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n = len(next(iter(c.values())))
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    e = {"launches": n, "GRBM_GUI_ACTIVE": gui, "mfma_util": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        e.update(valu_active_frac_of_wave_cycles=m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                 wait_inst_any_frac=m.get("SQ_WAIT_INST_ANY", 0.0) / wc, wait_any_frac=m.get("SQ_WAIT_ANY", 0.0) / wc)
    rows.append((n * gui, k, e))
for _, k, e in sorted(rows, reverse=True)[:14]:
    out["per_kernel"][k] = e
    print(f"{k[:66]:66s} n {e['launches']:4d} mfma_util {e['mfma_util']:.3f} valu {e.get('valu_active_frac_of_wave_cycles', 0):.3f} "
          f"wait_inst {e.get('wait_inst_any_frac', 0):.3f} wait_any {e.get('wait_any_frac', 0):.3f}")
import os
tag = os.environ.get("CRA5_PROF_TAG", "r03")
json.dump(out, open(f"gpurun_out/{tag}/{tag}_mfma_util.json", "w"), indent=1)
PY
