#!/usr/bin/env python3
"""How long does the HOST spend launching one captured training step (hipGraphLaunch of ~730 kernel nodes), against how long the GPU
takes to run it? If the two are equal the step is bound by the runtime's enqueue rate, not by the kernels.
usage: python tools/replay_host_time.py [reps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

bench = importlib.import_module("bench")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = bench.WORKLOADS["scratch"]
dev = "cuda:0"
torch.cuda.set_device(0)
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain(dev, "bf16", 1, wl["overrides"], wl["yaml"])
batch = batch_mod.synthetic_batch(wl["B"], wl["T"], wl["Te"], wl["U"], feats=True, seed=1234, enroll_emb_dim=wl["emb"]).to(dev)
brain.enable_hip_graph(warmup_steps=3)
for _ in range(6):
    brain.fit_batch(batch)
    torch.cuda.synchronize()
assert brain._graph is not None
for _ in range(5):
    brain.fit_batch(batch)
torch.cuda.synchronize()
print("single replays: host return / GPU done (ms after the call)")
for _ in range(6):
    t0 = time.perf_counter()
    brain.fit_batch(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"  host {1e3 * (t1 - t0):7.3f}   done {1e3 * (t2 - t0):7.3f}", flush=True)
t0 = time.perf_counter()
for _ in range(reps):
    brain.fit_batch(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{reps} replays back to back: host loop {1e3 * (t1 - t0) / reps:.3f} ms/step, until done {1e3 * (t2 - t0) / reps:.3f} ms/step")
