"""Uninitialised-read hunt: every torch.empty / empty_like / new_empty on the GPU comes back filled with NaN (floats) or 0xFF bytes
(workspaces); a training step whose result depends on memory it never wrote then shows NaN / a changed loss. Eager mode."""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_model_gpu as t
entry = t.entry
POISON = [False]
_empty, _empty_like = torch.empty, torch.empty_like
def _fill(x):
    if POISON[0] and x.is_cuda and x.numel():
        if x.dtype in (torch.float32, torch.bfloat16, torch.float16, torch.float64): x.fill_(float("nan"))
        elif x.dtype == torch.uint8: x.fill_(0xFF)
        elif x.dtype in (torch.int32, torch.int64): x.fill_(0x7F7F7F7F)
    return x
torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))
_new_empty = torch.Tensor.new_empty
torch.Tensor.new_empty = lambda self, *a, **k: _fill(_new_empty(self, *a, **k))

def run(poison, which):
    inp = t.golden_inputs()
    short = {k: (v[:, : v.shape[1] * 3 // 4] if k in ("mixed_sig", "enroll_sig") else v) for k, v in inp.items()}
    brain, h = entry._config1_brain(t.DEV, "bf16")
    brain.modules.train()
    batches = [t.make_batch(inp).to(t.DEV), t.make_batch(short).to(t.DEV)]
    POISON[0] = poison
    out = []
    for i in range(4):
        out.append(float(brain.fit_batch(batches[which if which >= 0 else i % 2])))
    POISON[0] = False
    names = []
    return out, brain

for which in (0, 1, -1):
    a, _ = run(False, which)
    b, brain = run(True, which)
    print("batch", which, "clean   ", a)
    print("batch", which, "poisoned", b, "MATCH" if a == b else "DIFF", flush=True)
