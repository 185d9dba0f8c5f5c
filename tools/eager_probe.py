#!/usr/bin/env python3
"""Like tools/eager_repeat.py, with device-side checksums of the forward's hand-over points (speaker embedding, predictor output,
encoder output) recorded on the stream that produced them (no host sync); for a run that differs from run 0 the first deviating
checksum tells which branch went wrong. usage: python tools/eager_probe.py [runs] [accum]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
entry = importlib.import_module("__graft_entry__")
tsasr = importlib.import_module("ts-asr_amd.recipes.tsasr")
from oracle.golden_recipe import golden_inputs
from test_model_gpu import make_batch
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
accum = int(sys.argv[2]) if len(sys.argv) > 2 else 2
inp = golden_inputs()
LOG = []
_pred, _spk = tsasr.TSASR._predictor, tsasr.TSASR._speaker_embedding


def pred(self, *a):
    out = _pred(self, *a)
    LOG.append(("dec", out.detach().float().abs().sum()))
    return out


def spk(self, *a):
    out, lens = _spk(self, *a)
    LOG.append(("spk", out.detach().float().abs().sum()))
    return out, lens


tsasr.TSASR._predictor, tsasr.TSASR._speaker_embedding = pred, spk


def one():
    del LOG[:]
    brain, h = entry._config1_brain("cuda", "bf16")
    brain.grad_accumulation_factor = accum
    brain.modules.train()
    ep = brain.modules.encoder_proj
    ep.register_forward_hook(lambda m, i, o: LOG.append(("enc", o.detach().float().abs().sum())))
    batch = make_batch(inp).to("cuda")
    for _ in range(8):
        LOG.append(("loss", brain.fit_batch(batch).clone()))
    torch.cuda.synchronize()
    return [(k, float(v)) for k, v in LOG]


ref, bad = one(), 0
for r in range(1, runs):
    got = one()
    if got != ref:
        bad += 1
        dev = [(i, k, a, b) for i, ((k, a), (_, b)) in enumerate(zip(got, ref)) if a != b]
        print(f"run {r}: first deviations (entry index, what, got, expected): {dev[:5]}   [{len(dev)} of {len(ref)} entries differ; 4 entries per step: spk, dec, enc, loss]")
print(f"{bad} of {runs - 1} repeats differ (accum {accum})")
