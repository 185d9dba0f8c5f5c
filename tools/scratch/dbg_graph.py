import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
if os.environ.get("GRAPH", "1") == "1":
    brain.enable_hip_graph(warmup_steps=3)
for i in range(9):
    loss = brain.fit_batch(batch)
    torch.cuda.synchronize()
    bad = [n for n, p in brain.modules.named_parameters() if not torch.isfinite(p).all()]
    print(i, float(loss), int(brain._nonfinite_dev.item()), float(brain.optimizer.last_grad_norm), bad[:6], len(bad), flush=True)
