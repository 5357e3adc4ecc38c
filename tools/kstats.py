"""Print a rocprofv3 *_kernel_stats.csv compactly: python tools/kstats.py DIR_OR_FILE [max_rows]"""
import csv
import glob
import os
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for i, r in enumerate(csv.reader(open(p))):
    if r[0] == "Name" or i > n:
        continue
    print(f"{r[0][:100]:100s} calls {r[1]:>6s} avg_us {float(r[3]) / 1000:9.1f} min {float(r[5]) / 1000:8.1f} tot_ms {float(r[2]) / 1e6:8.1f}")
