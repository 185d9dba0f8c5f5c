"""Which Python lines launch the ATen kernels of one eager training step (torch.profiler with stacks; scratch)."""
import importlib, os, sys
sys.path.insert(0, ".")
import torch
import bench
from torch.profiler import profile, ProfilerActivity
wl = dict(bench.WORKLOADS["scratch"])
bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U = wl["B"], wl["T"], wl["Te"], wl["U"]
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, wl["overrides"], wl["yaml"])
batch_mod = importlib.import_module(bench.PKG + ".batch")
batch = batch_mod.synthetic_batch(wl["B"], wl["T"], wl["Te"], wl["U"], feats=True, seed=1234, ragged=False, enroll_emb_dim=wl["emb"]).to("cuda:0")
for _ in range(3):
    brain.fit_batch(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    brain.fit_batch(batch)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.kernels:
        ks = [k.name for k in ev.kernels]
        if not any(("at::" in k) or ("rocclr" in k) or ("Memcpy" in k) or ("Memset" in k) for k in ks):
            continue
        st = [s for s in (ev.stack or []) if "site-packages" not in s and "dist-packages" not in s][:4]
        rows.append((ev.time_range.start, ev.name, ks[0][:50], tuple(ev.input_shapes or ()), st))
rows.sort()
seen = {}
for t, name, k, shp, st in rows:
    print(f"{name:12s} {k[17:48]:32s} " + " <- ".join(s.split("/")[-1][:60] for s in st))
print(len(rows), "ATen launches")
