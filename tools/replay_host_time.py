#!/usr/bin/env python3
"""Host time of one replay of the captured step (graph launch is asynchronous: if the host needs longer than the GPU, the step is
host-bound) with TSASR_GRAPH_SEGMENTS=0/1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, overrides=None)
import importlib
batch_mod = importlib.import_module(bench.PKG + ".batch")
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.enable_hip_graph(warmup_steps=3)
for _ in range(6):
    brain.fit_batch(batch)
torch.cuda.synchronize()
g = list(brain._graphs.values())[0]
for name in ("replay only",):
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(type(g).__name__, "host ms per replay / until GPU done:", " ".join("%.2f/%.2f" % t for t in ts[3:]))
if hasattr(g, "items"):
    for name, gg, st, deps in g.items:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(st):
            gg.replay()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("  %-24s host %.3f ms, GPU done after %.3f ms" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
