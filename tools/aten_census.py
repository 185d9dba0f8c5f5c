#!/usr/bin/env python3
"""Which PyTorch (library) ops still launch kernels inside one training step: per-op counts and GPU time (torch.profiler, eager)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
for _ in range(3):
    brain.fit_batch(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    brain.fit_batch(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) > 0]
rows.sort(key=lambda e: -e.count)
print(f"{'op':40s} {'calls':>6s} {'gpu us':>10s}")
for e in rows[:45]:
    t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    print(f"{e.key:40s} {e.count:6d} {t:10.0f}")
