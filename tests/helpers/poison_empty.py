#!/usr/bin/env python3
"""Does any kernel's result depend on uninitialised memory? Every torch.empty / empty_like on the GPU is filled with NaN (0xFF bytes
for integer workspaces) before use, then the eager training step (configs[0] model, ragged batch) must give bit-identical losses and
gradients to the unpoisoned run. usage: python tests/helpers/poison_empty.py [accum]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
entry = importlib.import_module("__graft_entry__")
from oracle.golden_recipe import golden_inputs
from test_model_gpu import make_batch
accum = int(sys.argv[1]) if len(sys.argv) > 1 else 2
inp = golden_inputs()
_empty, _empty_like = torch.empty, torch.empty_like
POISON = [False]


def _poison(t):
    if POISON[0] and t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(255)
        elif t.dtype in (torch.int32, torch.int64, torch.int16):
            t.fill_(-1)
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))


def run(poison):
    POISON[0] = False
    brain, h = entry._config1_brain("cuda", "bf16")
    brain.grad_accumulation_factor = accum
    brain.modules.train()
    batch = make_batch(inp).to("cuda")
    POISON[0] = poison
    out = []
    for _ in range(6):
        loss = brain.fit_batch(batch)
        torch.cuda.synchronize()
        out.append((float(loss), {n: None if p.grad is None else p.grad.detach().clone() for n, p in brain.modules.named_parameters()}))
    POISON[0] = False
    return out


clean, dirty = run(False), run(True)
ok = True
for i, ((la, ga), (lb, gb)) in enumerate(zip(clean, dirty)):
    dg = [n for n in ga if ga[n] is not None and not torch.equal(ga[n], gb[n])]
    same = (la == lb) and not dg
    ok &= same
    print(f"step {i}: loss clean {la!r} poisoned {lb!r}; gradients that differ: {len(dg)} {dg[:8]}")
print("OK: nothing reads uninitialised memory" if ok else "MISMATCH: some kernel depends on uninitialised memory")
sys.exit(0 if ok else 1)
