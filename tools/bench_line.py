"""Print the headline numbers of a bench.py JSON line.  usage: bench_line.py FILE [tag]  (never reads stdin)."""
import json
import sys

path = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else path
d = json.loads(open(path).read().strip().split("\n")[-1])
print(tag, "inflight", d["config"].get("frames_in_flight_per_gpu"), "fps", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 1),
      "gemm TF", round(d["roofline"]["achieved"], 1), "frac", round(d["roofline"]["frac"], 3), "gemm ms", round(d["roofline"]["gemm_ms_per_step"], 1),
      "attn ms", round(d["attention"]["ms_per_step"], 1))
