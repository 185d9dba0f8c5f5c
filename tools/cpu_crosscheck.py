#!/usr/bin/env python3
"""BASELINE.md section 3 cross-check, run in the 8-core BUILD container only (no GPU; /root/reference must exist): what bench.py's
`cpu_baseline` (kind "port" = oracle/tsasr_ref.py, the CPU restatement) is worth relative to the REFERENCE ITSELF.

In ONE process, back to back, on identical inputs, identical weights and identical exclusions:
  (R) the reference's own modules, imported from /root/reference exactly as oracle/gen_golden.py imports them (the three missing packages
      registered as empty modules: SURVEY.md section 8c), built with conformer-t_scratch.yaml's constructor arguments at FULL size
      (12 layers, d_model 256, d_ffn 2048, joint 640) and chained as train_librispeechmix_scratch.py:34-148 chains them;
  (O) oracle.tsasr_ref.compute_forward on the reference modules' state_dict.
Workload (the one BASELINE.md section 2 timed): B = 4, T = 1000 normalised mel frames, U = 120, fp32, 8 threads, train mode with
dropout 0, speaker branch and RNN-T loss EXCLUDED, `logits.log_softmax(-1).mean().backward()` drives backward. First the two are checked
to agree (logits and a gradient), then timed alternately (median of `steps` each).

usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/cpu_crosscheck.py [steps] [--out profiles/r05_cpu_crosscheck.json]"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gen_golden  # noqa: E402  (test infrastructure: the reference import recipe and the module constructors)
from oracle import tsasr_ref  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if args else 3
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
THREADS = 8
torch.set_num_threads(THREADS)
torch.manual_seed(0)
gen_golden.import_reference()
CFG = dict(gen_golden.CFG1, d_model=256, encoder_num_layers=12, speaker_num_layers=6, d_ffn=2048, joint_dim=640, decoder_neurons=512)
B, T, U = 4, 1000, 120
g = torch.Generator().manual_seed(999)
feats = torch.randn(B, T, CFG["n_mels"], generator=g)
lens = torch.ones(B)
tokens = torch.randint(1, CFG["vocab_size"], (B, U), generator=g)
tokens_bos = torch.cat([torch.zeros(B, 1, dtype=torch.long), tokens], 1)
tb_lens = torch.ones(B)

m = gen_golden.build(CFG, "cat", False, "same")           # deterministic weights (oracle/golden_recipe.py), eval() -> switch to train below
keep = ("frontend", "encoder", "encoder_proj", "embedding", "decoder", "decoder_proj", "joiner", "transducer_head")
for k in keep:
    m[k].train()                                          # dropout is 0 in these constructors: train mode = the training arithmetic
params = [p for k in keep for p in m[k].parameters() if p.requires_grad]


def reference_step():
    f = m["frontend"](feats)
    e = m["encoder"](f, lens, None, None)                # no speaker embedding: the injection is skipped (models/conformer.py:211-238)
    e = m["encoder_proj"](e)
    d, _ = m["decoder"](m["embedding"](tokens_bos), lengths=tb_lens)
    d = m["decoder_proj"](d)
    logits = m["transducer_head"](m["joiner"](e[..., None, :], d[:, None, ...]))
    logits.log_softmax(-1).mean().backward()
    return logits


sd = {f"{n}.{k}": v.detach().clone().requires_grad_(v.dtype.is_floating_point) for n in keep for k, v in m[n].state_dict().items()}
cfg_o = dict(nhead=CFG["nhead"], encoder_num_layers=CFG["encoder_num_layers"], speaker_num_layers=CFG["speaker_num_layers"],
             vocab_size=CFG["vocab_size"], blank_index=CFG["blank_index"])
cb = {"mixed_feats": feats, "mixed_lens": lens, "tokens_bos": tokens_bos, "tokens_bos_lens": tb_lens}


def oracle_step():
    logits = tsasr_ref.compute_forward(cb, sd, cfg_o, None, from_feats=True)
    logits.log_softmax(-1).mean().backward()
    return logits


# ---- same results first
lr = reference_step().detach()
gr = m["encoder"].layers[3].ffn_module1[1].ffn[0].weight.grad.clone()
lo = oracle_step().detach()
go = sd["encoder.layers.3.ffn_module1.1.ffn.0.weight"].grad.clone()
agree = {"logits_rel_l2": float((lo - lr).norm() / lr.norm()), "grad_rel_l2": float((go - gr).norm() / gr.norm())}
print("agreement:", agree, flush=True)
assert agree["logits_rel_l2"] < 1e-4 and agree["grad_rel_l2"] < 1e-3, agree


def clear():
    for p in params:
        p.grad = None
    for v in sd.values():
        v.grad = None


tr, to = [], []
for i in range(steps):
    for name, fn, acc in (("reference", reference_step, tr), ("oracle", oracle_step, to)):
        clear()
        t0 = time.perf_counter()
        fn()
        acc.append(time.perf_counter() - t0)
        print(f"  {name} step {i}: {acc[-1]:.2f} s", flush=True)
med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731
fr, fo = B * T / med(tr), B * T / med(to)
cpu_model = ""
try:
    with open("/proc/cpuinfo") as f:
        cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
except OSError:
    pass
out = {
    "workload": "fwd+bwd, conformer-t_scratch full size, B=4, T=1000 mel frames (normalised features in), U=120, fp32, dropout 0, no speaker "
                "branch, log_softmax(-1).mean() stand-in for the RNN-T loss; reference and oracle alternate in one process",
    "threads": THREADS, "cpu_count": os.cpu_count(), "cpu_model": cpu_model or platform.processor(), "steps_each": steps,
    "reference_frames_per_s": round(fr, 1), "oracle_frames_per_s": round(fo, 1), "reference_ratio": round(fo / fr, 3),
    "reference_step_s": [round(x, 2) for x in tr], "oracle_step_s": [round(x, 2) for x in to],
    "agreement": agree, "survey_figure_BASELINE_md": 412.0,
}
print(json.dumps(out))
if out_path:
    with open(out_path if os.path.isabs(out_path) else os.path.join(ROOT, out_path), "w") as f:
        json.dump(out, f, indent=1)
