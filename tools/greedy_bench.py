#!/usr/bin/env python3
"""Greedy transducer search at the benchmark's shapes (B = 32, T' = 250, random-init configs[1] model): the one-launch device decoder
(csrc/search.hip) against the per-frame host loop over library kernels (decoders.py, TSASR_GREEDY_KERNEL=0)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bench = importlib.import_module("bench")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
brain.modules.eval()
searcher = h["greedy_searcher"]
torch.manual_seed(0)
enc = (torch.randn(32, 250, h["joint_dim"], device="cuda") * 2).bfloat16()
res = {}
for mode in ("1", "0"):
    os.environ["TSASR_GREEDY_KERNEL"] = mode
    searcher(enc); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        hyps, score, _, _ = searcher(enc)
    torch.cuda.synchronize()
    res[mode] = ((time.time() - t0) / 3 * 1e3, hyps)
agree = sum(a == b for a, b in zip(res["1"][1], res["0"][1]))
print("device decoder %.2f ms, host loop %.1f ms per batch of 32 x 250 frames; %d symbols emitted; %d / 32 hypotheses identical (bf16 near-ties)" %
      (res["1"][0], res["0"][0], sum(len(x) for x in res["1"][1]), agree))
