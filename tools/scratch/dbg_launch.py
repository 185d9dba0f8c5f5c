import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.enable_hip_graph(warmup_steps=3)
for i in range(6):
    brain.fit_batch(batch)
torch.cuda.synchronize()
g = brain._graphs["step"]
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("replay() host call %.2f ms; until GPU done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
# back-to-back replays
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 replays: host %.2f ms total, GPU done %.2f ms -> %.2f ms/step" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 100))
