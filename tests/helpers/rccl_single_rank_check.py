#!/usr/bin/env python3
"""Exercise the multi-rank code path with REAL RCCL on one GPU: a one-rank communicator made through the C-ABI (csrc/comm.hip; RCCL
refuses two ranks on one device, so the two-rank rehearsal on one GPU uses gloo), the gradient arena told there are two ranks so that every
collective of the step is issued: bucketed all-reduces from the gradient hooks / the grouped weight-gradient flushes in the eager warm-up, then the SAME
collectives captured inside the step's hipGraph (RCCL kernels as graph nodes on the communication stream, joined before the fused
optimizer) and replayed; last run: the form a failed capture probe falls back to (dp.direct_capture_probe: nothing captured, one
all-reduce of the arena between replay and optimizer). AVG over one rank is the identity, so losses must equal the single-process run (to 1e-4: with bucket sends
the parameter-gradient reductions run per bucket instead of batched). Third run: bf16 all-reduce payload (TSASR_ALLREDUCE_DTYPE).
N > 1 ranks are NOT validated by this (no multi-GPU box in the development loop): it proves capture + replay of real RCCL nodes."""
import importlib
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29544"), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
import numpy as np
import torch
bench = importlib.import_module("bench")
dp = importlib.import_module(bench.PKG + ".dp")
batch_mod = importlib.import_module(bench.PKG + ".batch")
ops = importlib.import_module(bench.PKG + ".ops")
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
losses, sent = {}, {}
for mode in ("plain", "rccl", "rccl_bf16", "rccl_uncaptured"):
    torch.manual_seed(0)
    ops._seed_dev.clear()          # the device-side dropout step counter is process-global: restart it for every run
    brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, overrides=None)
    if mode != "plain":
        assert dp.direct_rccl_init(1, 0, "cuda:0", force=True) == 1   # a one-rank RCCL communicator through the C-ABI (csrc/comm.hip)
        brain.distributed = True
        brain.on_fit_start()
        brain.arena.world_size, brain.arena.direct = 2, True       # issue every collective (one-rank AVG = identity)
        brain.arena.comm_dtype = "bf16" if mode == "rccl_bf16" else "fp32"
    if mode == "rccl_uncaptured":
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29611"), rank=0, world_size=1)
        brain._graph_comm_ok = False     # what a failed capture probe (dp.direct_capture_probe) leaves: one all-reduce after the replay
    brain.enable_hip_graph(warmup_steps=3)
    ls = []
    for i in range(8):
        ls.append(float(brain.fit_batch(batch)))
    torch.cuda.synchronize()
    losses[mode] = ls
    sent[mode] = len(brain.arena.sent_log) if mode != "plain" else 0      # bucket collectives issued by the last traced step (the capture)
    if mode in ("rccl", "rccl_bf16"):
        assert brain._graph_comm_ok is True, "the captured all-reduce probe must pass on a working RCCL"
    print(mode, ["%.4f" % v for v in ls], "graphs:", len(brain._graphs), "bucket collectives in the captured step:", sent[mode], flush=True)
    assert len(brain._graphs) == 1
np.testing.assert_allclose(losses["rccl"], losses["plain"], rtol=1e-4)   # (gradient sums are ordered differently when buckets are sent during backward)
np.testing.assert_allclose(losses["rccl_bf16"], losses["plain"], rtol=2e-2)
np.testing.assert_allclose(losses["rccl_uncaptured"], losses["plain"], rtol=1e-4)
assert sent["rccl_uncaptured"] == 0, sent
assert sent["rccl"] >= 6 and sent["rccl_bf16"] >= 6, sent
C = importlib.import_module(bench.PKG + "._capi")
C.check(C.lib().tsasr_allreduce_destroy(), "tsasr_allreduce_destroy")
torch.distributed.destroy_process_group()
print("RCCL single-rank path OK")
