"""Runs the multi-shape graph==eager test N times in one process and counts mismatches (numerical flake hunt, not a fault hunt)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_model_gpu as t
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for i in range(n):
    try:
        t.test_hip_graph_cache_per_batch_shape()
    except AssertionError as e:
        bad += 1
        print("run", i, "MISMATCH", str(e).splitlines()[3:6], flush=True)
print("mismatches", bad, "of", n, flush=True)
