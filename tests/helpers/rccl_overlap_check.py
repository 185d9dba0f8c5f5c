#!/usr/bin/env python3
"""The overlap PROPERTY of the bucketed gradient all-reduce (SB/core.py:1585-1615 wraps the modules in DistributedDataParallel, whose reducer
overlaps bucket all-reduces with backward; here dp.GradArena does), checked on one GPU with real RCCL collectives on a one-rank communicator
(the arena is told there are two ranks, so every collective is issued): with small buckets (TSASR_BUCKET_MB=8: ~26 of them)
  (1) the first bucket's collective is ISSUED while backward is still running (host order), and all but the last few buckets are;
  (2) an event recorded behind that collective on the communication stream completes BEFORE the event that marks the end of backward on
      the main stream (device order: the collective did not wait for backward to finish);
  (3) the same step captured into a hipGraph issues the same collectives in the same order (sent_log) and replays to the eager losses.
Not its speed: one rank moves no bytes over xGMI."""
import importlib
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29547"), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
os.environ["TSASR_BUCKET_MB"] = "8"
import numpy as np
import torch
bench = importlib.import_module("bench")
dp = importlib.import_module(bench.PKG + ".dp")
batch_mod = importlib.import_module(bench.PKG + ".batch")
ops = importlib.import_module(bench.PKG + ".ops")
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
torch.manual_seed(0)
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, overrides=None)
assert dp.direct_rccl_init(1, 0, "cuda:0", force=True) == 1
brain.distributed = True
brain.on_fit_start()
brain.arena.world_size, brain.arena.direct = 2, True
eager = []
for i in range(3):                       # eager steps: the bucket order settles in the first one
    eager.append(float(brain.fit_batch(batch)))
nb = len(brain.arena.buckets)
assert nb >= 6, nb
brain.arena.probe = {}
eager.append(float(brain.fit_batch(batch)))
torch.cuda.synchronize()
pr, order_eager = brain.arena.probe, list(brain.arena.sent_log)
brain.arena.probe = None
assert pr["first_in_backward"], "the first bucket's collective must be issued during backward"
assert pr["sent_during_backward"] >= nb - 3, (pr["sent_during_backward"], nb)
lead_ms = pr["first_done"].elapsed_time(pr["backward_done"])
assert lead_ms > 0.0, lead_ms            # the first collective had completed before backward ended
print(f"{nb} buckets; {pr['sent_during_backward']} collectives issued during backward; the first one completed {lead_ms:.3f} ms before backward ended", flush=True)
brain.enable_hip_graph(warmup_steps=0)
graph = [float(brain.fit_batch(batch)) for _ in range(4)]
torch.cuda.synchronize()
assert brain._graph_comm_ok is True and len(brain._graphs) == 1
assert list(brain.arena.sent_log) == order_eager and len(order_eager) == nb      # the captured step issued the same collectives in the same order
assert all(np.isfinite(eager + graph))
C = importlib.import_module(bench.PKG + "._capi")
C.check(C.lib().tsasr_allreduce_destroy(), "tsasr_allreduce_destroy")
print("RCCL overlap property OK")
