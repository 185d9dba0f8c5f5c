#!/usr/bin/env python3
"""Which PyTorch (library) ops still launch kernels inside one training step, and from which source line (torch.profiler, eager)."""
import collections, importlib, os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    brain.fit_batch(batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    src = next((f for f in (e.stack or []) if "ts-asr_amd" in f or "bench.py" in f), None)
    if src is None:   # backward ops run on the autograd thread: attribute them to the autograd node that issued them
        par = e.cpu_parent
        while par is not None and not (par.name.endswith("Backward") or "Backward" in par.name or par.name.startswith("autograd::")):
            par = par.cpu_parent
        src = "(bwd) " + par.name if par is not None else "(other)"
    src = src.split("ts-asr_amd/")[-1][:70]
    agg[(e.name, src)][0] += 1
    agg[(e.name, src)][1] += t
print(f"{'op':28s} {'calls':>5s} {'gpu us':>8s}  source")
for (name, src), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{name:28s} {n:5d} {t:8.0f}  {src}")
