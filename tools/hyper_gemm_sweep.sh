#!/bin/bash
# per-shape kernel durations (median of 30) of every small-GEMM instantiation + the big engine
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "1 1" "1 2" "1 4" "1 8" "2 1" "2 2" "2 4" "4 1" "big x"; do
  d=/tmp/hy_$(echo $v | tr ' ' '_'); rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python $R/tools/hyper_gemm_sweep.py $v > /dev/null 2>&1
  python - "$v" $d <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "split_rows" not in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# 7 shapes x 30 launches in order
out = []
for i in range(7):
    seg = d[i * 30:(i + 1) * 30]
    out.append(statistics.median(seg) if seg else float("nan"))
print(f"{sys.argv[1]:6s}: " + "  ".join(f"{x:6.1f}" for x in out))
PY
done
echo "shapes: 648x1080x360 648x360x360 648x1440x360 648x360x1440 648x360x4096 648x8192x360 648x256x360"
