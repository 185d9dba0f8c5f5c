#!/usr/bin/env python3
"""BASELINE.md section 3 cross-check, run in the 8-core BUILD container (no GPU needed): the oracle (oracle/tsasr_ref.py, the CPU
restatement that bench.py times as `cpu_baseline`) on the workload BASELINE.md section 2 timed the REFERENCE's own modules on -
full conformer-t_scratch sizes, B = 4, T = 1000 mel frames, U = 120, fp32, 8 threads, speaker branch and RNN-T loss EXCLUDED
(a log_softmax().mean() stand-in drives backward, as there) - should land within +-15 % of that 412 frames/s figure.
usage: python tools/cpu_crosscheck.py [steps]"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import tsasr_ref  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.set_num_threads(8)
hp = importlib.import_module("ts-asr_amd.hparams")
batch_mod = importlib.import_module("ts-asr_amd.batch")
with open(os.path.join(ROOT, "hparams", "conformer-t_scratch_mi355x.yaml")) as f:
    h = hp.load_hyperpyyaml(f, dict(input_is_feats=True))
sd = {f"{n}.{k}": v.detach().float().clone().requires_grad_(v.dtype.is_floating_point)
      for n, m in h["modules"].items() if isinstance(m, torch.nn.Module) and not n.startswith("speaker") for k, v in m.state_dict().items()}
cfg = dict(nhead=4, encoder_num_layers=12, speaker_num_layers=6, vocab_size=29, blank_index=0)
B, T, U = 4, 1000, 120
bt = batch_mod.synthetic_batch(B, T, 500, U, feats=True, seed=999)
cb = {"mixed_feats": bt.mixed_sig.data, "mixed_lens": bt.mixed_sig.lengths, "tokens_bos": bt.tokens_bos.data, "tokens_bos_lens": bt.tokens_bos.lengths}
times = []
for i in range(1 + steps):
    t0 = time.perf_counter()
    logits = tsasr_ref.compute_forward(cb, sd, cfg, None, from_feats=True)
    logits.log_softmax(-1).mean().backward()
    for v in sd.values():
        v.grad = None
    dt = time.perf_counter() - t0
    print(f"step {i}: {dt:.2f} s", flush=True)
    if i > 0:
        times.append(dt)
med = sorted(times)[len(times) // 2]
out = {"frames_per_s": round(B * T / med, 1), "threads": 8, "cpu_count": os.cpu_count(), "steps": steps, "reference_figure_BASELINE_md": 412.0,
       "ratio": round(B * T / med / 412.0, 3), "workload": "fwd+bwd, B=4, T=1000, U=120, fp32, no speaker branch, log_softmax().mean() stand-in for the loss"}
print(json.dumps(out))
