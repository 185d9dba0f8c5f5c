#!/usr/bin/env python3
"""Exercise the multi-rank code path with the REAL RCCL backend on one GPU: a one-rank "nccl" process group (RCCL refuses two ranks
on one device, so tools/ddp_rehearsal.sh uses gloo), the gradient arena told there are two ranks so that every collective of the
step is issued (bucketed all-reduce from the gradient hooks in the eager warm-up, hipGraph capture with the RCCL watchdog thread
alive, all-reduce between the captured step and the optimizer). AVG over one rank is the identity, so losses must equal the
single-process run (to 1e-4: with bucket sends from the hooks the parameter-gradient reductions run one by one instead of batched)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
import torch
bench = importlib.import_module("bench")
dp = importlib.import_module(bench.PKG + ".dp")
batch_mod = importlib.import_module(bench.PKG + ".batch")
ops = importlib.import_module(bench.PKG + ".ops")
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
losses = {}
for mode in ("plain", "rccl"):
    torch.manual_seed(0)
    ops._seed_dev.clear()          # the device-side dropout step counter is process-global: restart it for the second run
    if mode == "rccl":
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, overrides=None)
    if mode == "rccl":
        brain.distributed = True
        brain.on_fit_start()
        brain.arena.world_size, brain.arena.group = 2, None      # issue every collective (one-rank AVG = identity)
    brain.enable_hip_graph(warmup_steps=3)
    ls = []
    for i in range(8):
        ls.append(float(brain.fit_batch(batch)))
    torch.cuda.synchronize()
    losses[mode] = ls
    print(mode, ["%.4f" % v for v in ls], "graphs:", len(brain._graphs), flush=True)
import numpy as np
np.testing.assert_allclose(losses["rccl"], losses["plain"], rtol=1e-4)   # (gradient sums are ordered differently when buckets are sent from the hooks)
print("RCCL single-rank path OK")
torch.distributed.destroy_process_group()
