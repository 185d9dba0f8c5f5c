#!/usr/bin/env python3
"""Does any kernel's result depend on what the PREVIOUS kernel left in LDS? The eager training step (configs[0] model, ragged batch) is
run with every C-ABI launch preceded by a fill of all LDS with a pattern (NaN bits, then a large finite value); losses and gradients
must be bit-identical to the plain run. usage: python tests/helpers/lds_garbage.py [accum]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
entry = importlib.import_module("__graft_entry__")
C = importlib.import_module("ts-asr_amd._capi")
from oracle.golden_recipe import golden_inputs
from test_model_gpu import make_batch
accum = int(sys.argv[1]) if len(sys.argv) > 1 else 2
inp = golden_inputs()
lib = C.lib()
PATTERN = [None]
fill = C.lab().tsasr_lab_fill_lds      # lab equipment (include/tsasr_lab.h), not part of the product ABI
SKIP = ("tsasr_last_error", "tsasr_version", "tsasr_device_ok")


class Wrapped:
    """ctypes library proxy: every tsasr_* entry point that takes a stream launches the LDS fill on that stream first."""
    def __init__(self, real):
        self._real, self._cache = real, {}

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if name in SKIP or not name.startswith("tsasr_") or "workspace" in name or "bytes" in name:
            return f
        if name not in self._cache:
            def call(*a, _f=f):
                if PATTERN[0] is not None and torch.cuda.is_available():
                    fill(PATTERN[0], C.stream_ptr())
                return _f(*a)
            self._cache[name] = call
        return self._cache[name]


proxy = Wrapped(lib)
C.lib = lambda: proxy


def run(pattern):
    PATTERN[0] = None
    brain, h = entry._config1_brain("cuda", "bf16")
    brain.grad_accumulation_factor = accum
    brain.modules.train()
    batch = make_batch(inp).to("cuda")
    PATTERN[0] = pattern
    out = []
    for _ in range(4):
        loss = brain.fit_batch(batch)
        torch.cuda.synchronize()
        out.append((float(loss), {n: None if p.grad is None else p.grad.detach().clone() for n, p in brain.modules.named_parameters()}))
    PATTERN[0] = None
    return out


clean = run(None)
ok = True
for pat, label in ((0xFFFFFFFF, "NaN bits"), (0x7F7F7F7F, "3.4e38"), (0x3F803F80, "ones (bf16 pairs)")):
    dirty = run(pat)
    for i, ((la, ga), (lb, gb)) in enumerate(zip(clean, dirty)):
        dg = [n for n in ga if ga[n] is not None and not torch.equal(ga[n], gb[n])]
        if la != lb or dg:
            ok = False
            print(f"LDS = {label}, step {i}: loss {la!r} -> {lb!r}; gradients that differ: {len(dg)} {dg[:10]}")
print("OK: no kernel depends on leftover LDS contents" if ok else "MISMATCH: some kernel reads LDS it did not write")
sys.exit(0 if ok else 1)
