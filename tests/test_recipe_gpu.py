"""The runnable recipe (train_tsasr.py: parse_arguments -> load_hyperpyyaml -> TSASR -> Brain.fit -> Brain.evaluate, the counterpart of
train_librispeechmix_scratch.py:491-611 / SB/core.py:1288-1381,1492-1563) on synthetic batches, and run-to-run reproducibility of the
captured training step (gradients and weights bit for bit: no float atomics, fixed reduction orders)."""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
SMALL = ["--d_model", "144", "--nhead", "4", "--encoder_num_layers", "2", "--speaker_num_layers", "2", "--d_ffn", "576", "--joint_dim", "160",
         "--decoder_neurons", "128", "--compute_dtype", "bf16"]


@pytest.mark.parametrize("yaml_name,graph", [("conformer-t_scratch_mi355x.yaml", True), ("conformer-t_wavlm_mi355x.yaml", False),
                                             ("conformer-t_none_mi355x.yaml", True)])
def test_recipe_main_fit_and_evaluate(yaml_name, graph):
    """Two epochs of Brain.fit over 6 length-bucketed synthetic batches (validation with the greedy searcher after each), then
    Brain.evaluate with the beam searcher: losses finite and falling, hypotheses produced, optimizer steps counted."""
    main = importlib.import_module("train_tsasr").main
    core = importlib.import_module("ts-asr_amd.core")
    argv = [os.path.join(ROOT, "hparams", yaml_name), "--device", "cuda:0", "--synthetic", "6", "--number_of_epochs", "2", "--syn_batch", "4",
            "--syn_seconds", "2.0", "--syn_enroll_seconds", "1.0", "--syn_tokens", "12", "--hip_graph", str(graph), "--lr", "0.002",
            "--warmup_steps", "5", "--dropout", "0.0", "--beam_size", "3"] + SMALL
    losses = []
    ops = importlib.import_module("ts-asr_amd.ops")
    ops.LIB_FALLBACKS.clear()
    orig = core.Brain.on_stage_end
    try:
        core.Brain.on_stage_end = lambda self, stage, loss, epoch=None: losses.append((stage, loss))
        brain, result = main(argv)
    finally:
        core.Brain.on_stage_end = orig
    train = [l for s, l in losses if s == core.Stage.TRAIN]
    valid = [l for s, l in losses if s == core.Stage.VALID]
    assert len(train) == 2 and len(valid) == 2 and all(np.isfinite(train + valid))
    assert train[1] < train[0]                                   # it trains
    assert result["optimizer_steps"] == 12 and result["nonfinite"] == 0
    assert np.isfinite(result["test_loss"])
    hyps = brain.last_hyps                                       # TEST stage: beam search hypotheses, one token list per utterance
    assert isinstance(hyps, list) and len(hyps) == 4 and all(isinstance(h, list) for h in hyps)
    if graph:
        assert len(brain._graphs) >= 2                           # one captured graph per batch shape (three length buckets)
    # fit + validation (greedy search) + evaluation (beam search) of a recipe YAML never left the hand-written kernels (ops.STATUS "LIB" rows)
    assert ops.LIB_FALLBACKS == {}, ops.LIB_FALLBACKS


def test_library_routes_are_counted_and_refused_in_strict_mode(monkeypatch):
    """The routes ops.STATUS lists as LIB: an LSTM shape no kernel takes (two layers) runs the library module, is counted and announced once;
    with ops.STRICT_HIP the same call raises TsasrHipMissing."""
    import torch
    ops = importlib.import_module("ts-asr_amd.ops")
    capi = importlib.import_module("ts-asr_amd._capi")
    rnn = torch.nn.LSTM(8, 16, num_layers=2, batch_first=True).to("cuda:0")
    x = torch.randn(2, 5, 8, device="cuda:0")
    ops.LIB_FALLBACKS.clear()
    with pytest.warns(RuntimeWarning, match="library kernel"):
        out, _ = ops.lstm(x, rnn)
    assert out.shape == (2, 5, 16) and ops.LIB_FALLBACKS == {"lstm": 1}
    monkeypatch.setattr(ops, "STRICT_HIP", True)
    with pytest.raises(capi.TsasrHipMissing):
        ops.lstm(x, rnn)
    ops.LIB_FALLBACKS.clear()


def _hash_arena(t):
    return hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()


def _run_steps(tag):
    """8 captured steps of the headline workload's model at BASELINE configs[1] shapes in a fresh child process: hashes of the
    gradients of a frozen-weight replay and of every weight after each optimizer step."""
    import subprocess
    code = r'''
import hashlib, importlib, os, sys
sys.path.insert(0, %r)
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.modules.train()
brain.enable_hip_graph(warmup_steps=2)
out = []
for it in range(7):
    loss = float(brain.fit_batch(batch))
    torch.cuda.synchronize()
    out.append("w%%d %%r %%s" %% (it, loss, hashlib.md5(brain.arena.flat_params.cpu().numpy().tobytes()).hexdigest()))
brain.grad_accumulation_factor = 10 ** 9          # accumulate-only micro-steps from now on: weights frozen, gradients of one replay
for it in range(3):
    brain.arena.zero_()
    loss = float(brain.fit_batch(batch))
    torch.cuda.synchronize()
    out.append("g%%d %%r %%s" %% (it, loss, hashlib.md5(brain.arena.grads.cpu().numpy().tobytes()).hexdigest()))
print("\n".join(out))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return [l for l in r.stdout.splitlines() if l[:1] in "wg" and l[1:2].isdigit()]


def test_captured_step_is_bitwise_reproducible_across_processes():
    """Weights after every optimizer step and the gradients of replayed steps are identical, bit for bit, between two processes
    (two forked HIP streams, grouped weight gradients, batched reductions; dropout on: the counter-based masks are a function of the
    step index). Round 1 chased one such mismatch to a packed-fp32 instruction; this keeps watch."""
    a, b = _run_steps("a"), _run_steps("b")
    assert len(a) == 10 and a == b, "\n".join(x + "   |   " + y for x, y in zip(a, b) if x != y)
    g = [l.split()[-1] for l in a if l.startswith("g")]
    assert len(set(g)) >= 2       # different dropout masks per replay: the gradients do change from step to step
