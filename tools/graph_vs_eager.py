#!/usr/bin/env python3
"""Which parameters differ between an eager run and a hipGraph-replayed run of the same steps (configs[0] model, dropout 0)?
Prints, per step, the parameters whose values are not bit-identical (name, max |diff|): a debugging aid for tests/test_model_gpu.py's
test_hip_graph_replay_equals_eager. usage: python tools/graph_vs_eager.py [steps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
entry = importlib.import_module("__graft_entry__")
from oracle.golden_recipe import golden_inputs
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import make_batch  # noqa
DEV = "cuda"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
inp = golden_inputs()
brains = {}
for mode in ("eager", "graph"):
    brain, h = entry._config1_brain(DEV, "bf16")
    brain.modules.train()
    if mode == "graph":
        brain.enable_hip_graph(warmup_steps=2)
    brains[mode] = (brain, make_batch(inp).to(DEV))
for it in range(steps):
    ls = {m: float(b.fit_batch(bt)) for m, (b, bt) in brains.items()}
    torch.cuda.synchronize()
    pe = dict(brains["eager"][0].modules.named_parameters()); pg = dict(brains["graph"][0].modules.named_parameters())
    bad = [(n, float((pe[n].float() - pg[n].float()).abs().max())) for n in pe if not torch.equal(pe[n], pg[n])]
    print(f"step {it}: loss eager {ls['eager']!r} graph {ls['graph']!r}; {len(bad)} / {len(pe)} parameters differ")
    for n, d in bad[:12]:
        print(f"    {n}: {d:.3e}")
    if bad:
        break
