"""GPU parity: fused joint+head and RNN-T loss HIP kernels (through the C-ABI) vs the CPU oracle."""
import importlib

import numpy as np
import pytest
import torch

from oracle import rnnt_ref as RR
from oracle import tsasr_ref as R
from oracle.golden_recipe import det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def rn():
    return importlib.import_module("ts-asr_amd.rnnt")


@pytest.fixture
def mfma_fp32(monkeypatch):
    """fp32 STORAGE through the MFMA joint kernels (h, W, dlogits rounded to bf16 inside) - the fp32 path proper is csrc/joint_f32.hip."""
    monkeypatch.setattr(importlib.import_module("ts-asr_amd.rnnt"), "JOINT_F32_EXACT", False)


def bf(x):
    return x.to(torch.bfloat16).float()


def joint_oracle(enc, dec, W, b, emulate_bf16, fp16_hidden=False):
    """oracle/tsasr_ref.joint_logits; with emulate_bf16 the two MFMA operands are rounded as the kernel rounds them. fp16_hidden: the
    bf16 TS-ASR joint (J = 640: csrc/rnnt.hip joint_fwd_regw_kernel) forms the hidden activation in packed fp16 - x = fp16(e + d),
    h = max(x, fp16(x * fp16(slope))) - and multiplies it with the fp16-rounded head matrix."""
    if not emulate_bf16:
        return R.joint_logits(enc, dec, {"w.weight": W, "w.bias": b}, "")
    if fp16_hidden:
        x = enc.half()[:, :, None, :] + dec.half()[:, None, :, :]
        h = torch.maximum(x, x * torch.tensor(0.01).half())
        return h.float() @ W.half().float().t() + b
    h = bf(torch.nn.functional.leaky_relu(enc[:, :, None, :] + dec[:, None, :, :], 0.01))
    return h @ bf(W).t() + b


def make(B, T, U1, J, V, seed, dtype=torch.float32):
    enc = torch.from_numpy(det_tensor(f"rnnt.enc{seed}", (B, T, J), 1.0))
    dec = torch.from_numpy(det_tensor(f"rnnt.dec{seed}", (B, U1, J), 1.0))
    W = torch.from_numpy(det_tensor(f"rnnt.W{seed}", (V, J), 1.0 / np.sqrt(J)))
    b = torch.from_numpy(det_tensor(f"rnnt.b{seed}", (V,), 0.1))
    tg = torch.from_numpy(np.random.default_rng(seed).integers(1, V, size=(B, U1 - 1)).astype(np.int32))
    if dtype == torch.bfloat16:
        enc, dec = bf(enc), bf(dec)
    return enc, dec, W, b, tg


@pytest.mark.parametrize("B,T,U1,J,V", [(4, 50, 21, 160, 29), (2, 7, 33, 64, 5), (1, 1, 1, 32, 32), (3, 130, 70, 640, 29)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joint_forward(rn, B, T, U1, J, V, dtype, mfma_fp32):
    enc, dec, W, b, _ = make(B, T, U1, J, V, 1, dtype)
    out = rn.fused_joint_logits(enc.to(DEV, dtype), dec.to(DEV, dtype), W.to(DEV), b.to(DEV))
    assert out.shape == (B, T, U1, V) and out.stride(-2) == 32
    ref = joint_oracle(enc, dec, W, b, True, fp16_hidden=(dtype == torch.bfloat16 and J == 640))
    # operands rounded identically -> only fp32 accumulation order differs
    torch.testing.assert_close(out.cpu(), ref, atol=2e-3, rtol=1e-3)
    # against the un-rounded fp32 oracle: bf16 operand rounding, |logit| ~ 1: 2^-8 relative per operand
    torch.testing.assert_close(out.cpu(), joint_oracle(enc, dec, W, b, False), atol=6e-2, rtol=2e-2)
    # padded columns are zero
    full = out.as_strided((B, T, U1, 32), (T * U1 * 32, U1 * 32, 32, 1))
    assert torch.all(full[..., V:] == 0)


def test_reference_known_answer_on_gpu(rn):
    """vendor/speechbrain/tests/unittests/test_losses.py:109-152: 2.2478 on the Numba (/T) path = 4.4957 default path."""
    lg = torch.tensor([[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
                        [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]], device=DEV).requires_grad_()
    tg = torch.tensor([[1, 2]], device=DEV, dtype=torch.int32)
    one = torch.tensor([1.0], device=DEV)
    loss = rn.transducer_loss(lg, tg, one, one, blank_index=0, use_torchaudio=True)
    loss.backward()
    assert loss.item() == pytest.approx(4.4957, abs=2e-4)
    loss_numba = rn.transducer_loss(lg.detach().log_softmax(-1), tg, one, one, blank_index=0, use_torchaudio=False)
    assert loss_numba.item() == pytest.approx(2.2478, rel=1e-4)
    _, g = RR.transducer_loss_ref(lg.detach().cpu().numpy(), tg.cpu().numpy(), [1.0], [1.0], 0, "mean")
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g, atol=1e-6)


LOSS_CASES = [
    # B, T, U1, V, tlen, ulen
    (4, 50, 21, 29, [50, 45, 40, 35], [20, 18, 15, 10]),
    (3, 6, 5, 7, [6, 4, 1], [4, 2, 0]),          # ragged incl. T=1 and empty target
    (2, 9, 65, 11, [9, 3], [64, 63]),            # K=2 lanes, lattice wider than it is long
    (2, 40, 130, 29, [40, 17], [129, 100]),      # K=4
    (1, 300, 300, 29, [300], [299]),             # K=8
    (2, 33, 600, 6, [33, 20], [599, 1]),         # K=16
]


@pytest.mark.parametrize("B,T,U1,V,tlen,ulen", LOSS_CASES)
def test_loss_and_grad(rn, B, T, U1, V, tlen, ulen):
    rng = np.random.default_rng(B * 1000 + T)
    lg = (rng.standard_normal((B, T, U1, V)) * 2).astype(np.float32)
    tg = rng.integers(1, V, size=(B, U1 - 1)).astype(np.int32) if U1 > 1 else np.zeros((B, 1), np.int32)
    costs_ref, grads_ref = RR.rnnt_costs_grads(lg, tg, tlen, ulen, 0)
    x = torch.from_numpy(lg).to(DEV).requires_grad_()
    gsc = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(DEV)
    costs = rn.rnnt_costs(x, torch.from_numpy(tg).to(DEV), torch.tensor(tlen, device=DEV, dtype=torch.int32),
                          torch.tensor(ulen, device=DEV, dtype=torch.int32), 0)
    (costs * gsc).sum().backward()
    # fp32 log-space DP with fast exp/log: absolute error grows with the T+U chain; 1e-5 relative on costs ~ 1e2
    np.testing.assert_allclose(costs.detach().cpu().numpy(), costs_ref, rtol=2e-5, atol=1e-4)
    g = x.grad.cpu().numpy()
    # alpha/beta are fp32 sums of up to T+U terms of magnitude ~|cost|: the log-domain error, hence the RELATIVE error of
    # every occupancy exp(alpha+beta-logP), grows with the lattice perimeter (the oracle is float64)
    np.testing.assert_allclose(g, grads_ref * gsc.cpu().numpy()[:, None, None, None], atol=3e-5, rtol=max(1e-3, 5e-5 * (T + U1)))
    for b in range(B):  # exactly zero outside the lattice
        assert np.all(g[b, tlen[b]:] == 0) and np.all(g[b, :, ulen[b] + 1:] == 0)


PLAN_CASES = [
    (3, 70, 520, 7, [70, 41, 9], [519, 300, 64]),       # K = 16: ragged in both directions, one utterance shorter than the 64-step ramp
    (1, 200, 1100, 5, [200], [1099]),                    # K = 32: the long-form layout (16 column blocks when split)
    (2, 5, 300, 6, [5, 1], [299, 0]),                    # K = 8: fewer frames than one prefetch window; an empty target
]


@pytest.mark.parametrize("B,T,U1,V,tlen,ulen", PLAN_CASES)
def test_lattice_plans_give_the_same_bits(rn, B, T, U1, V, tlen, ulen):
    """Every way csrc/rnnt.hip walks a long lattice - frame-major planes with 4 waves (round 3), anti-diagonal planes with 4 / 8 / 16 waves,
    one single-wave workgroup per block of 64 / 128 / 256 columns with the edge values handed from block to block through global memory -
    returns the same costs and gradients bit for bit, and they match the float64 oracle."""
    C = importlib.import_module("ts-asr_amd._capi")
    rng = np.random.default_rng(B * 977 + U1)
    lg = (rng.standard_normal((B, T, U1, V)) * 2).astype(np.float32)
    tg = rng.integers(1, V, size=(B, U1 - 1)).astype(np.int32)
    costs_ref, grads_ref = RR.rnnt_costs_grads(lg, tg, tlen, ulen, 0)
    plans = [(0, 4, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0), (2, 1, 0), (1, 8, 1), (1, 8, 2), (1, 8, 4), (-1, -1, -1)]
    outs = []
    try:
        for plan in plans:
            C.lib().tsasr_rnnt_lattice_plan(*plan)
            x = torch.from_numpy(lg).to(DEV).requires_grad_()
            costs = rn.rnnt_costs(x, torch.from_numpy(tg).to(DEV), torch.tensor(tlen, device=DEV, dtype=torch.int32),
                                  torch.tensor(ulen, device=DEV, dtype=torch.int32), 0)
            costs.sum().backward()
            outs.append((costs.detach().cpu().numpy(), x.grad.cpu().numpy()))
    finally:
        C.lib().tsasr_rnnt_lattice_plan(-1, -1, -1)
    np.testing.assert_allclose(outs[0][0], costs_ref, rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(outs[0][1], grads_ref, atol=3e-5, rtol=max(1e-3, 5e-5 * (T + U1)))
    for plan, (c, g) in zip(plans[1:], outs[1:]):
        assert np.array_equal(c, outs[0][0]), plan
        assert np.array_equal(g, outs[0][1]), plan


def test_loss_rejects_bad_arguments(rn):
    C = importlib.import_module("ts-asr_amd._capi")
    lg = torch.zeros(1, 2, 3, 5, device=DEV)
    one = torch.tensor([1], device=DEV, dtype=torch.int32)
    with pytest.raises(ValueError):
        rn.rnnt_costs(lg, torch.zeros(1, 1, device=DEV, dtype=torch.int32), one, one, 0)  # targets too short
    with pytest.raises(C.TsasrHipError):
        rn.rnnt_costs(lg, torch.zeros(1, 2, device=DEV, dtype=torch.int32), one, one, 7)  # blank outside vocabulary
    with pytest.raises(C.TsasrHipMissing):
        rn.rnnt_costs(lg.cpu(), torch.zeros(1, 2, dtype=torch.int32), one.cpu(), one.cpu(), 0)  # no CPU path


@pytest.mark.parametrize("B,T,U1,J,V,tlen,ulen", [(4, 50, 21, 160, 29, [50, 45, 40, 35], [20, 18, 15, 10]),
                                                   (2, 19, 40, 64, 9, [19, 8], [39, 3]),
                                                   (2, 70, 121, 640, 29, [70, 66], [120, 77])])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joint_loss_backward_chain(rn, B, T, U1, J, V, tlen, ulen, dtype, mfma_fp32):
    """enc/dec/head gradients of mean RNN-T loss through the fused joint: HIP vs oracle (autograd through the
    bf16-operand-emulating joint + C oracle loss)."""
    enc, dec, W, b, tg = make(B, T, U1, J, V, 3, dtype)
    # oracle
    eo, do_, Wo, bo = (t.clone().requires_grad_() for t in (enc, dec, W, b))
    h = torch.nn.functional.leaky_relu(eo[:, :, None, :] + do_[:, None, :, :], 0.01)
    h = h + (bf(h) - h).detach()  # straight-through bf16 rounding of the MFMA operand
    Wr = Wo + (bf(Wo) - Wo).detach()
    logits_o = h @ Wr.t() + bo
    tl_t, ul_t = torch.tensor(tlen, dtype=torch.int32), torch.tensor(ulen, dtype=torch.int32)
    loss_o = RR.RnntLossRefFn.apply(logits_o, tg, tl_t, ul_t, 0).mean()
    loss_o.backward()
    # HIP
    eg, dg = enc.to(DEV, dtype).requires_grad_(), dec.to(DEV, dtype).requires_grad_()
    Wg, bg = W.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    logits = rn.fused_joint_logits(eg, dg, Wg, bg, 0.01, tl_t.to(DEV), ul_t.to(DEV))
    loss = rn.rnnt_costs(logits, tg.to(DEV), tl_t.to(DEV), ul_t.to(DEV), 0).mean()
    loss.backward()
    assert loss.item() == pytest.approx(loss_o.item(), rel=1e-4)
    # gradients: dlogits are rounded to bf16 for the MFMA (2^-9 relative) and, in bf16 mode, outputs are bf16
    tol = dict(atol=3e-3, rtol=3e-2) if dtype == torch.float32 else dict(atol=6e-3, rtol=5e-2)
    scale = max(1.0, float(eo.grad.abs().max()))
    torch.testing.assert_close(eg.grad.float().cpu() / scale, eo.grad / scale, **tol)
    torch.testing.assert_close(dg.grad.float().cpu() / scale, do_.grad / scale, **tol)
    wscale = max(1.0, float(Wo.grad.abs().max()))
    torch.testing.assert_close(Wg.grad.cpu() / wscale, Wo.grad / wscale, **tol)
    torch.testing.assert_close(bg.grad.cpu(), bo.grad, atol=2e-3, rtol=2e-2)
    # relative L2 error is the sharper check
    for a, r in ((eg.grad, eo.grad), (dg.grad, do_.grad), (Wg.grad, Wo.grad)):
        rel = (a.float().cpu() - r).norm() / r.norm()
        assert rel < (6e-3 if dtype == torch.float32 else 1.2e-2), rel


@pytest.mark.parametrize("B,T,U1,J,V,tlen,ulen", [(4, 50, 21, 160, 29, [50, 45, 40, 35], [20, 18, 15, 10]),
                                                   (2, 19, 40, 64, 9, [19, 8], [39, 3]),
                                                   (2, 70, 121, 640, 29, [70, 66], [120, 77]),
                                                   (1, 3, 5, 36, 32, [3], [4])])
def test_joint_exact_fp32_chain(rn, B, T, U1, J, V, tlen, ulen):
    """The fp32 parity path (csrc/joint_f32.hip: no matrix cores, nothing rounded to bf16): logits against the UN-rounded oracle
    (oracle/tsasr_ref.joint_logits) to 1e-5, and enc / dec / head gradients of the mean RNN-T loss against autograd through that oracle +
    the C oracle loss to 5e-5 relative L2 (the MFMA kernels with fp32 storage are held to 6e-3 above)."""
    assert rn.JOINT_F32_EXACT
    enc, dec, W, b, tg = make(B, T, U1, J, V, 4, torch.float32)
    eo, do_, Wo, bo = (t.clone().requires_grad_() for t in (enc, dec, W, b))
    logits_o = R.joint_logits(eo, do_, {"w.weight": Wo, "w.bias": bo}, "")
    tl_t, ul_t = torch.tensor(tlen, dtype=torch.int32), torch.tensor(ulen, dtype=torch.int32)
    loss_o = RR.RnntLossRefFn.apply(logits_o, tg, tl_t, ul_t, 0).mean()
    loss_o.backward()
    eg, dg = enc.to(DEV).requires_grad_(), dec.to(DEV).requires_grad_()
    Wg, bg = W.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    logits = rn.fused_joint_logits(eg, dg, Wg, bg, 0.01, tl_t.to(DEV), ul_t.to(DEV))
    torch.testing.assert_close(logits.detach().cpu(), logits_o.detach(), atol=1e-5, rtol=1e-5)
    full = logits.as_strided((B, T, U1, 32), (T * U1 * 32, U1 * 32, 32, 1))
    assert torch.all(full[..., V:] == 0)
    loss = rn.rnnt_costs(logits, tg.to(DEV), tl_t.to(DEV), ul_t.to(DEV), 0).mean()
    loss.backward()
    assert loss.item() == pytest.approx(loss_o.item(), rel=2e-6)
    for a, r, name in ((eg.grad, eo.grad, "denc"), (dg.grad, do_.grad, "ddec"), (Wg.grad, Wo.grad, "dW"), (bg.grad, bo.grad, "dbias")):
        rel = float((a.float().cpu() - r).norm() / r.norm())
        assert rel < 5e-5, (name, rel)     # fp32 sums over T * U1 lattice cells (measured 2.7e-5 on denc)


def test_full_size_properties(rn):
    """BASELINE.json configs[1] lattice (B=32, T'=250, U+1=121, V=29, J=640): loss equals the C oracle on the HIP logits;
    sum over v of dlogits rows is 0; grad is zero outside ragged lattices; determinism (bitwise) of two runs."""
    B, T, U1, J, V = 32, 250, 121, 640, 29
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(B, T, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
    dec = torch.randn(B, U1, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
    W = (torch.randn(V, J, generator=g) / J ** 0.5).to(DEV).requires_grad_()
    b = torch.zeros(V, device=DEV).requires_grad_()
    tg = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32)
    tl = torch.linspace(0.6, 1.0, B).mul(T).round().int()
    ul = torch.linspace(0.5, 1.0, B).mul(U1 - 1).round().int()
    outs = []
    for _ in range(2):
        for p in (enc, dec, W, b):
            p.grad = None
        logits = rn.fused_joint_logits(enc, dec, W, b, 0.01, tl.to(DEV), ul.to(DEV))
        logits.retain_grad()
        costs = rn.rnnt_costs(logits, tg.to(DEV), tl.to(DEV), ul.to(DEV), 0)
        costs.mean().backward()
        outs.append((costs.detach().clone(), logits.grad.clone(), enc.grad.clone(), dec.grad.clone(), W.grad.clone()))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)  # no atomics anywhere -> bitwise reproducible
    lg = logits.detach().cpu().numpy()
    costs_ref, grads_ref = RR.rnnt_costs_grads(np.ascontiguousarray(lg), tg.numpy(), tl.numpy(), ul.numpy(), 0)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), costs_ref, rtol=2e-5)
    dl = outs[0][1].cpu().numpy()
    np.testing.assert_allclose(dl, grads_ref / B, atol=2e-6, rtol=1e-3)
    np.testing.assert_allclose(dl.sum(-1), 0, atol=5e-6)
