"""Do two deferred gradient adds of one step target the same parameter (two workgroups of accumulate_many on one destination)?"""
import collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.modules.train()
brain.fit_batch(batch)
names = {id(p): n for n, p in brain.modules.named_parameters()}
orig = brain.arena._flush_deferred
def traced():
    c = collections.Counter(id(p) for _, p in brain.arena._deferred)
    print("deferred adds:", len(brain.arena._deferred), "distinct params:", len(c))
    for k, v in c.items():
        if v > 1:
            print("  DUPLICATE", names.get(k), v, [tuple(g.shape) for g, p in brain.arena._deferred if id(p) == k])
    # overlapping source / destination ranges
    iv = sorted((p.grad.data_ptr(), p.grad.data_ptr() + 4 * p.numel(), names.get(id(p))) for _, p in brain.arena._deferred)
    for a, b in zip(iv, iv[1:]):
        if b[0] < a[1]:
            print("  OVERLAP dst", a[2], b[2])
    return orig()
brain.arena._flush_deferred = traced
brain.fit_batch(batch)
torch.cuda.synchronize()
