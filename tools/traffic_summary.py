"""Per-kernel HBM-side traffic from the two rocprofv3 PMC passes of tools/profile_bench.sh
(FETCH_SIZE and WRITE_SIZE, each collected in its own run with --kernel-trace only).

   python tools/traffic_summary.py gpurun_out profiles/r01     -> profiles/r01_traffic.json,
                                                                  profiles/r01_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read -> doubled; both counters are in KB; counted at the L2<->fabric boundary
(Infinity-Cache hits included), i.e. an upper bound on HBM bytes."""
import collections
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
extra = (" " + sys.argv[3]) if len(sys.argv) > 3 else ""   # e.g. "--precision f16"
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}/pmc_traffic_{c}/pmc_counter_collection.csv")):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    rows = sorted(((k, len(v), sum(v) / len(v), sum(v)) for k, v in agg.items()), key=lambda t: -t[3])
    with open(f"{dst}_pmc_{c}_per_kernel.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Dispatches", f"Avg_{c}_KB", "Total_KB"])
        for k, n, a, t in rows[:40]:
            w.writerow([k, n, round(a, 1), round(t, 1)])
    per[c] = {k: (n, a) for k, n, a, t in rows}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of "
                 f"`python bench.py{extra} --steps 1 --warmup 1 --exclusive --inflight 1 --no-kernel-timer` on 1 x MI355X "
                 "(tools/profile_bench.sh, tools/traffic_summary.py)",
       "correction": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM "
                     "section); KB -> bytes; counted at the L2<->fabric boundary, Infinity-Cache hits included: an "
                     "upper bound on HBM bytes.",
       "per_kernel": {}}
tot_b, tot_n = 0.0, 0
for k, (n, f) in per["FETCH_SIZE"].items():
    wv = per["WRITE_SIZE"].get(k, (n, 0.0))[1]
    b = (2.0 * f + wv) * 1024.0
    short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
    out["per_kernel"][short] = {"launches": n, "bytes_per_launch": b, "fetch_bytes": 2.0 * f * 1024.0, "write_bytes": wv * 1024.0}
    if "gemm_nt_split_kernel<2, 4," in k:
        tot_b += b * n
        tot_n += n
out["gemm_nt_split"] = {"launches": tot_n, "avg_bytes_per_launch": tot_b / max(tot_n, 1),
                        "kernels": "gemm_nt_split_kernel<2,4,{3|4},2> (the 192x256 / 256x256 tile instantiations)"}
json.dump(out, open(f"{dst}_traffic.json", "w"), indent=1)
print(json.dumps(out["gemm_nt_split"]))
