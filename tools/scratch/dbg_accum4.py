import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 4)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.enable_hip_graph(warmup_steps=3)
for i in range(12):
    brain.fit_batch(batch)
torch.cuda.synchronize()
ga, gs = brain._graphs["accumulate"], brain._graphs["step"]
def t(g, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("accumulate flavour alone: %.2f ms" % t(ga))
print("step flavour alone:       %.2f ms" % t(gs))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ga.replay(); ga.replay(); ga.replay(); gs.replay()
torch.cuda.synchronize(); print("a,a,a,s pattern: %.2f ms per micro-batch" % ((time.perf_counter() - t0) / 20 * 1e3))
