#!/usr/bin/env python3
"""Pretty-print the per-kernel-family HIP-event timings of a bench.py JSON line (stdin or file)."""
import json, sys
d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
k = d["hip_kernels"]
steps = 3 if d["config"].get("hip_graph") else d["steps"]
tot = 0.0
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    t = v["avg_ms"] * v["launches"] / steps
    tot += t
    print("%7.3f ms/step %5d x %8.1f us  %-52s %s" % (t, v["launches"] // steps, v["avg_ms"] * 1e3, n, v.get("TFLOPps", "")))
print("sum of instrumented regions %.2f ms/step; bench %.2f ms/step, %.0f frames/s" % (tot, d["ms_per_step"], d["value"]))
print("roofline:", d["roofline"])
