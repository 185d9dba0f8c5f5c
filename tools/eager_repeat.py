#!/usr/bin/env python3
"""Run-to-run stability of the EAGER training step without per-step synchronisation (the host runs ahead of the GPU, as in
tests/test_model_gpu.py test_hip_graph_replay_equals_eager): N fresh brains, 8 steps each on the same batch; prints every loss
sequence that differs from the first. usage: python tools/eager_repeat.py [runs] [accum]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
entry = importlib.import_module("__graft_entry__")
from oracle.golden_recipe import golden_inputs
from test_model_gpu import make_batch
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
accum = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inp = golden_inputs()
first, bad = None, 0
for r in range(runs):
    brain, h = entry._config1_brain("cuda", "bf16")
    brain.grad_accumulation_factor = accum
    brain.modules.train()
    batch = make_batch(inp).to("cuda")
    ls = [float(brain.fit_batch(batch)) for _ in range(8)]
    if first is None:
        first = ls
    elif ls != first:
        bad += 1
        print(f"run {r}: differs from run 0 at steps {[i for i, (a, b) in enumerate(zip(ls, first)) if a != b]}: {[round(x, 4) for x in ls]}")
print(f"{bad} of {runs - 1} repeats differ (accum {accum}); run 0: {[round(x, 4) for x in first]}")
