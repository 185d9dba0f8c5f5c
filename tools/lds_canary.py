#!/usr/bin/env python3
"""Does any kernel of the training step write LDS outside its own allocation? A canary kernel (tsasr_debug_lds_canary: workgroups that
hold a pattern in LDS and keep verifying it) runs on one HIP stream while whole training steps - every hand-written kernel of the path,
forward and backward - run on another. Corrupted canary words = some kernel's LDS write (an LDS-DMA, most likely) landed in a
neighbouring workgroup's allocation on the same CU.
usage: python tools/lds_canary.py [config1|bench] [steps] [canary_lds_bytes] [canary_wgs]"""
import ctypes
import importlib
import os
import sys

os.environ.setdefault("TSASR_OVERLAP", "0")       # the step itself on ONE stream: only the canary runs beside it
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "config1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lds_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
wgs = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
C = importlib.import_module("ts-asr_amd._capi")
if which == "config1":
    entry = importlib.import_module("__graft_entry__")
    from oracle.golden_recipe import golden_inputs
    from test_model_gpu import make_batch
    brain, h = entry._config1_brain("cuda", "bf16")
    batch = make_batch(golden_inputs()).to("cuda")
else:
    bench = importlib.import_module("bench")
    batch_mod = importlib.import_module("ts-asr_amd.batch")
    brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
    batch = batch_mod.synthetic_batch(32, 1000, 500, 120, feats=True, seed=1234).to("cuda:0")
brain.modules.train()
for _ in range(3):
    brain.fit_batch(batch)
torch.cuda.synchronize()
errors = torch.zeros(1, dtype=torch.int32, device="cuda")
first = torch.zeros(4, dtype=torch.int32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3000
for it in range(steps):
    C.check(C.lib().tsasr_debug_lds_canary(wgs, lds_bytes, iters, C.ptr(errors), C.ptr(first), ctypes.c_void_p(sb.cuda_stream)), "canary")
    with torch.cuda.stream(sa):
        brain.fit_batch(batch)
    if it % 20 == 19:
        torch.cuda.synchronize()
        print(f"  step {it + 1}: {int(errors.item())} corrupted canary words so far; first: {[hex(v & 0xffffffff) for v in first.tolist()]}", flush=True)
torch.cuda.synchronize()
n = int(errors.item())
print(f"RESULT {which}: {n} corrupted canary words in {steps} steps (canary: {wgs} workgroups x {lds_bytes} B of LDS, {iters} checks each);"
      f" first {{workgroup, word, read, expected}} = {[hex(v & 0xffffffff) for v in first.tolist()]}")
# the canary alone (nothing beside it) must stay clean
errors.zero_()
for _ in range(20):
    C.check(C.lib().tsasr_debug_lds_canary(wgs, lds_bytes, iters, C.ptr(errors), C.ptr(first), ctypes.c_void_p(sb.cuda_stream)), "canary")
torch.cuda.synchronize()
print(f"canary alone: {int(errors.item())} corrupted words")
sys.exit(1 if n else 0)
