#!/usr/bin/env python3
"""Which host calls issue device-to-device memcpy kernels (__amd_rocclr_copyBuffer) inside one eager training step (torch.profiler)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    brain.fit_batch(batch)
    torch.cuda.synchronize()
agg = collections.Counter()
names = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        names[e.name[:60]] += 1
    n = e.name.lower()
    if not ("memcpy" in n or "copybuffer" in n or "memset" in n or "fillbuffer" in n):
        continue
    par, chain = e.cpu_parent if hasattr(e, "cpu_parent") else None, []
    agg[e.name[:50]] += 1
print("device events by name:")
for k, v in names.most_common(25):
    print(f"  {v:5d}  {k}")
# CPU-side ops that launched a memcpy: aten::copy_ / aten::clone / aten::contiguous with their python stack
ops_ = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like") and e.device_type == torch.autograd.DeviceType.CPU:
        kids = [k.name for k in (e.kernels or [])] if hasattr(e, "kernels") else []
        if not any("copyBuffer" in k or "Memcpy" in k or "fillBuffer" in k or "Memset" in k for k in kids):
            continue
        src = next((f for f in (e.stack or []) if "ts-asr_amd" in f or "bench.py" in f), None)
        if src is None:
            par = e.cpu_parent
            while par is not None and "Backward" not in par.name and not par.name.startswith("autograd::"):
                par = par.cpu_parent
            src = "(bwd) " + (par.name if par is not None else "?")
        ops_[(e.name, src.split("ts-asr_amd/")[-1][:80], ",".join(sorted(set(k[:24] for k in kids))))] += 1
print("host ops that launched a memcpy / memset kernel:")
for (n, src, kk), c in ops_.most_common(40):
    print(f"  {c:4d}  {n:18s} {src}   [{kk}]")
