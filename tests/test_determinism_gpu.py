"""Run-to-run stability of the training step (round-2 verdict item 1): fresh brains, two batch shapes alternating (length-bucketed
batches: one captured hipGraph per shape, shared memory pool), gradient accumulation 1 and 2, eager and hipGraph - every loss of every
repeat must equal the first eager run's BIT FOR BIT (the reference trains on one deterministic stream: SB/core.py:1032-1096).
tools/det_stress.py is the long form of this test (thousands of repeats, per-module probes); profiles/r03_notes.md has what it found."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
entry = importlib.import_module("__graft_entry__")


def _run(mode, accum, steps, batches):
    brain, h = entry._config1_brain(DEV, "bf16")
    brain.grad_accumulation_factor = accum
    brain.modules.train()
    if mode == "graph":
        brain.enable_hip_graph(warmup_steps=2)
    out = [float(brain.fit_batch(batches[i % len(batches)])) for i in range(steps)]
    if mode == "graph":
        assert len(brain._graphs) >= len(batches)      # one graph per (shape, flavour) that occurred: here every shape meets one flavour
    return out


@pytest.mark.parametrize("accum", [1, 2])
def test_step_is_bit_stable_over_repeats_two_shapes_eager_and_graph(accum):
    from oracle.golden_recipe import golden_inputs
    from test_model_gpu import make_batch
    inp = golden_inputs()
    short = {k: (v[:, : v.shape[1] * 3 // 4] if k in ("mixed_sig", "enroll_sig") else v) for k, v in inp.items()}
    batches = [make_batch(inp).to(DEV), make_batch(short).to(DEV)]
    steps = 12 if accum == 1 else 16
    ref = _run("eager", accum, steps, batches)
    assert all(torch.isfinite(torch.tensor(ref)))
    bad = []
    for rep in range(15):
        for mode in ("eager", "graph"):
            got = _run(mode, accum, steps, batches)
            if got != ref:
                bad.append((rep, mode, [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]))
    assert not bad, f"runs that differ from the first eager run (repeat, mode, steps): {bad}"
