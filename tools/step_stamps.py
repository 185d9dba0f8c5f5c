#!/usr/bin/env python3
"""When does each phase of the captured training step REALLY start? TSASR_STAMPS=1 makes the recipe drop one-thread kernels that store the
device wall clock at phase boundaries (prof.stamp), on whatever stream is current there; they are captured with the step and replayed
with it, so the times below come from an ordinary, unprofiled replay (rocprofv3's kernel trace shows a more serial schedule than the
one that runs without it). usage: python tools/step_stamps.py [replays to show] [--config scratch|none|pretrained|longform]"""
import importlib
import os
import sys

os.environ["TSASR_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

bench = importlib.import_module("bench")
prof = importlib.import_module(bench.PKG + ".prof")
cfg = "scratch"
if "--config" in sys.argv:
    i = sys.argv.index("--config")
    cfg = sys.argv[i + 1]
    del sys.argv[i:i + 2]
shows = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = bench.WORKLOADS[cfg]
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
for _ in range(10):
    brain.fit_batch(batch)
torch.cuda.synchronize()
runs = []
for _ in range(shows):
    brain.fit_batch(batch)
    torch.cuda.synchronize()
    runs.append(prof.stamps_us())
names = [n for n, _ in runs[0]]
order = sorted(range(len(names)), key=lambda i: runs[-1][i][1])
print("microseconds after the step's first kernel, one column per replay:")
for i in order:
    print("  " + "  ".join(f"{r[i][1]:9.1f}" for r in runs) + "   " + names[i])
