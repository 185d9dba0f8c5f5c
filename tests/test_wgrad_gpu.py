"""GPU parity of the grouped weight-gradient launch (csrc/wgrad.hip) through the C-ABI: every queued dW += dy^T . x against the
fp32 product of the same bf16 operands (the reference's autograd computes exactly that product per weight: SB/nnet/linear.py:64-78
via torch.nn.functional.linear's backward), bit-exact repeatability, ragged token counts, edge tiles, accumulate semantics."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    o = importlib.import_module("ts-asr_amd.ops")
    o.reduce_defer_prepare(torch.device(DEV))
    return o


def _jobs(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (M, N, K) in shapes:   # dW [M, N] += dy [K, M]^T . x [K, N]
        dy = torch.randn(K, M, generator=g).to(torch.bfloat16)
        x = torch.randn(K, N, generator=g).to(torch.bfloat16)
        w0 = torch.randn(M, N, generator=g)
        out.append((dy, x, w0))
    return out


SHAPES = [
    [(2048, 256, 8000), (256, 2048, 8000), (768, 256, 8000), (256, 256, 8000)],                # the mixture encoder's layer
    [(2048, 256, 4000), (512, 256, 4000), (640, 256, 8000), (640, 512, 3872), (256, 2560, 8000)],   # ragged tokens (4000 = 62.5 k-tiles), 640-wide edges
    [(8, 8, 1), (24, 40, 63), (264, 256, 65), (256, 264, 128), (136, 520, 200)],                # tiny / odd: one k-tile with a tail, edge tiles both ways
]


@pytest.mark.parametrize("shapes", SHAPES)
def test_grouped_wgrad_vs_fp32_product(ops, shapes):
    jobs = _jobs(shapes, 7 + len(shapes))
    dev = [(dy.to(DEV), x.to(DEV), w0.clone().to(DEV)) for dy, x, w0 in jobs]
    params = [torch.nn.Parameter(torch.empty(0)) for _ in dev]
    for p, (dy, x, w) in zip(params, dev):
        assert ops.wgrad_queue(p, w, dy, x)
    assert ops.wgrad_pending() == len(dev)
    done = ops.wgrad_flush()
    assert len(done) == len(dev) and ops.wgrad_pending() == 0
    torch.cuda.synchronize()
    for (dy, x, w0), (_, _, w) in zip(jobs, dev):
        ref = w0.double() + dy.double().t() @ x.double()
        K = dy.shape[0]
        np.testing.assert_allclose(w.cpu().numpy(), ref.numpy(), atol=3e-4 * max(K, 16) ** 0.5, rtol=2e-5)
    # bit-exact repeatability (fixed summation order, exclusive tiles) and accumulation on top of the previous result
    again = [(dy, x, w0.clone().to(DEV)) for (dy, x, _), (_, _, w0) in zip(dev, jobs)]
    for p, (dy, x, w) in zip(params, again):
        assert ops.wgrad_queue(p, w, dy, x)
    ops.wgrad_flush()
    torch.cuda.synchronize()
    for (_, _, w), (_, _, w2) in zip(dev, again):
        assert torch.equal(w, w2)


def test_same_weight_twice_is_ordered(ops):
    """Two gradients into the same weight (a shared Linear) must not race: the second queue call flushes the first."""
    (dy, x, w0), (dy2, x2, _) = _jobs([(256, 256, 500), (256, 256, 300)], 3)
    p = torch.nn.Parameter(torch.empty(0))
    w = w0.clone().to(DEV)
    a, b, c, d = dy.to(DEV), x.to(DEV), dy2.to(DEV), x2.to(DEV)
    assert ops.wgrad_queue(p, w, a, b)
    assert ops.wgrad_queue(p, w, c, d)
    assert ops.wgrad_pending() == 1          # the first one already ran
    ops.wgrad_flush()
    ref = w0.double() + dy.double().t() @ x.double() + dy2.double().t() @ x2.double()
    np.testing.assert_allclose(w.cpu().numpy(), ref.numpy(), atol=2e-2, rtol=2e-5)


def test_strided_operands_and_rejects(ops):
    """Row-strided views (the conv front-end's centre tap, a slice of the arena) are taken; shapes the kernel cannot tile are refused
    (the caller then runs the per-weight GEMM)."""
    g = torch.Generator().manual_seed(11)
    big = torch.randn(700, 9 * 128, generator=g).to(torch.bfloat16).to(DEV)
    x = big[:, 4 * 128:5 * 128]                     # [700, 128], row stride 1152
    dy = torch.randn(700, 128, generator=g).to(torch.bfloat16).to(DEV)
    arena = torch.zeros(128 * 128 + 64, device=DEV)
    w = arena[64:].view(128, 128)
    p = torch.nn.Parameter(torch.empty(0))
    assert ops.wgrad_queue(p, w, dy, x)
    ops.wgrad_flush()
    ref = dy.double().t().cpu() @ x.double().cpu()
    np.testing.assert_allclose(w.cpu().numpy(), ref.numpy(), atol=2e-2, rtol=2e-5)
    assert not ops.wgrad_queue(p, torch.zeros(29, 640, device=DEV), torch.zeros(100, 29, device=DEV, dtype=torch.bfloat16),
                               torch.zeros(100, 640, device=DEV, dtype=torch.bfloat16))     # 29 rows: not a multiple of 8
    assert not ops.wgrad_queue(p, torch.zeros(32, 64, device=DEV), torch.zeros(100, 32, device=DEV), torch.zeros(100, 64, device=DEV))  # fp32 operands
    assert ops.wgrad_pending() == 0
