import sys, json
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(tag, "inflight", d["config"].get("frames_in_flight_per_gpu"), "fps", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 1),
      "gemm TF", round(d["roofline"]["achieved"], 1), "gemm ms", round(d["roofline"]["gemm_ms_per_step"], 1),
      "attn ms", round(d["attention"]["ms_per_step"], 1))
