"""Multi-rank machinery on the one-GPU box. Real RCCL: a one-rank "nccl" group with the arena told there are two ranks, so that the
bucketed all-reduces are issued during backward, CAPTURED into the step's hipGraph and replayed (tests/helpers/rccl_single_rank_check.py, run
as a child process: a process group per run, never a re-exec of a process that touched the GPU). N > 1 is covered by the gloo tests
of tests/test_host_cpu.py and stays unmeasured on hardware until a multi-GPU node runs bench.py --gpus N."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def test_rccl_collectives_captured_in_graph_and_replayed():
    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_single_rank_check.py")], env=env, capture_output=True, text=True, timeout=420)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "RCCL single-rank path OK" in r.stdout


def test_bucket_collectives_are_issued_and_complete_during_backward():
    """Overlap property of the bucketed all-reduce on one GPU (tests/helpers/rccl_overlap_check.py): with ~26 buckets of 8 MiB the first
    collective is issued while backward runs and completes before backward ends; the captured step issues the same collectives in order."""
    env = dict(os.environ, MASTER_PORT="29548")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_overlap_check.py")], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "RCCL overlap property OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_on_one_gpu_over_gloo_equal_single_process(graph):
    """Data parallel with device tensors: two ranks share the GPU, gloo carries the bucket all-reduces (tests/helpers/dp_gloo_gpu_check.py):
    ranks end with identical weights, equal to one process on the whole batch up to bf16 rounding - eagerly (bucket collectives from the
    gradient hooks, overlapped with backward) and with the step captured (one all-reduce between replay and optimizer: gloo cannot be
    captured). RCCL / xGMI are not exercised."""
    env = dict(os.environ, MASTER_PORT=str(29900 + os.getpid() % 90 + (7 if graph else 0)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "dp_gloo_gpu_check.py")] + (["--graph"] if graph else []), env=env,
                       capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "two-rank data parallel on one GPU (gloo) OK" in r.stdout


# ---- N > 1 on real hardware: these arm themselves the day the box has two GPUs (none has had so far: SCALE_r01-r03 were skipped) --------
_TWO_GPUS = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 visible MI355X (one rank per GPU over RCCL / xGMI)")


@_TWO_GPUS
@pytest.mark.parametrize("graph", [False, True])
def test_two_gpus_direct_rccl_equal_single_process(graph):
    """One rank per GPU, bucket all-reduces through the direct RCCL communicator (csrc/comm.hip; replaces the reference's per-module
    DDP reducers, SB/core.py:1464-1484, SB/utils/distributed.py:123-201): ranks end bit-identical, equal to ONE process on the whole batch
    to bf16 rounding - eagerly (collectives from the gradient hooks, overlapped with backward) and with the step captured into a
    hipGraph that CARRIES the bucket collectives (graph_comm on both ranks)."""
    env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 90 + (11 if graph else 0)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "dp_gloo_gpu_check.py"), "--rccl"] + (["--graph"] if graph else []),
                       env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "two-rank data parallel on two GPUs (direct RCCL) OK" in r.stdout


@_TWO_GPUS
def test_bench_gpus2_reports_direct_rccl_in_the_graph():
    """`python bench.py --gpus 2` (it spawns its own ranks): the JSON line must say the gradient path ran on a 2-rank direct RCCL
    communicator with the collectives inside the captured step - not a gloo rehearsal, not an all-reduce outside the graph."""
    import json
    env = dict(os.environ, MASTER_PORT=str(29800 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TSASR_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "4", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64
    comm = line["comm"]
    assert comm["backend"] == "nccl" and comm["rccl_direct_ranks"] == 2 and comm["collectives_in_graph"] and not comm["rehearsal_on_one_gpu"], comm
    assert comm["allreduce_busbw_GBps"] > 0


def test_speed_perturbation_varies_under_hip_graph():
    """`augment: True` + enable_hip_graph(): the speed SpeedPerturb picks per batch (speech_augmentation.py:480-493) must keep varying -
    drawn on the host before each replay, one graph per (batch shape, speed) - and the run must equal the eager run step for step."""
    import __graft_entry__ as entry
    from oracle.golden_recipe import golden_inputs
    from tests.test_variants_gpu import make_batch
    inp = golden_inputs()
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(77)                  # the speed draws come from torch's CPU generator, as in the reference
        importlib.import_module("ts-asr_amd.ops")._seed_dev.clear()
        brain, h = entry._config1_brain(DEV, "bf16", augment=True)
        brain.modules.train()
        if mode == "graph":
            brain.enable_hip_graph(warmup_steps=2)
        batch = make_batch(inp).to(DEV)
        seq, ls = [], []
        for _ in range(14):
            ls.append(float(brain.fit_batch(batch)))
            seq.append(brain.modules.speed_perturb.samp_index)
        runs[mode] = (seq, ls, len(brain._graphs))
    seq, ls, ngraphs = runs["graph"]
    assert len(set(seq)) == 3, seq                      # all three speeds (95 / 100 / 105 %) were used
    assert seq == runs["eager"][0]                      # the same draws as the eager run
    assert ngraphs == 3                                 # one captured graph per speed
    np.testing.assert_allclose(ls, runs["eager"][1], rtol=2e-5)
