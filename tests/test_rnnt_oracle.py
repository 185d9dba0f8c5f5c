"""Pins the RNN-T loss oracle (oracle/rnnt_ref.c): the reference's known-answer test, brute-force path
enumeration in float64, alpha/beta consistency, and gradients vs autograd through a log-space DP."""
import numpy as np
import pytest
import torch

from oracle import rnnt_ref as RR

# vendor/speechbrain/tests/unittests/test_losses.py:120-134 (data of the reference's known-answer test)
KAT_LOGITS = np.array([[[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
                        [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]], np.float32)
KAT_TARGETS = np.array([[1, 2]], np.int32)


def test_reference_known_answer():
    """The reference asserts 2.2478 +- 1e-4 on its Numba path, which divides by T (=2)
    (speechbrain/nnet/loss/transducer_loss.py:104-106); the default torchaudio path does not."""
    loss, _ = RR.transducer_loss_ref(KAT_LOGITS, KAT_TARGETS, [1.0], [1.0], 0, "mean")
    assert loss == pytest.approx(4.4957, abs=2e-4)
    assert loss / 2 == pytest.approx(2.2478, rel=1e-4)
    # log_softmax applied first (as the reference test does) changes nothing
    ls = torch.from_numpy(KAT_LOGITS).log_softmax(-1).numpy()
    loss2, _ = RR.transducer_loss_ref(ls, KAT_TARGETS, [1.0], [1.0], 0, "mean")
    assert loss2 == pytest.approx(loss, abs=1e-6)


@pytest.mark.parametrize("T,U,V", [(1, 0, 3), (1, 2, 4), (3, 0, 4), (2, 2, 5), (4, 3, 6), (5, 4, 29)])
def test_brute_force(T, U, V):
    rng = np.random.default_rng(T * 100 + U * 10 + V)
    lg = rng.standard_normal((1, T, U + 1, V)).astype(np.float32) * 2
    tg = rng.integers(1, V, size=(1, max(U, 1))).astype(np.int32)
    costs, _ = RR.rnnt_costs_grads(lg, tg, [T], [U], 0)
    assert costs[0] == pytest.approx(RR.brute_force_cost(lg[0], tg[0], T, U, 0), rel=1e-9, abs=1e-9)


def _torch_dp(logits, targets, T, U, blank=0):
    lp = logits.double().log_softmax(-1)
    a = [[None] * (U + 1) for _ in range(T)]
    for t in range(T):
        for u in range(U + 1):
            if t == 0 and u == 0:
                a[t][u] = lp.new_zeros(())
                continue
            terms = []
            if t > 0:
                terms.append(a[t - 1][u] + lp[t - 1, u, blank])
            if u > 0:
                terms.append(a[t][u - 1] + lp[t, u - 1, targets[u - 1]])
            a[t][u] = torch.logsumexp(torch.stack(terms), 0)
    return -(a[T - 1][U] + lp[T - 1, U, blank])


def test_gradients_and_ragged_batch():
    rng = np.random.default_rng(7)
    B, T, U1, V, ldl = 3, 6, 5, 7, 8  # padded rows (ldl > V) as the product's logits buffer has them
    lg = rng.standard_normal((B, T, U1, ldl)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, U1 - 1)).astype(np.int32)
    tl, ul = np.array([6, 4, 1], np.int32), np.array([4, 2, 0], np.int32)
    costs, grads, al, be = RR.rnnt_costs_grads(lg, tg, tl, ul, 0, V=V, want_ab=True)
    for b in range(B):
        x = torch.from_numpy(lg[b, :, :, :V]).clone().requires_grad_(True)
        c = _torch_dp(x, tg[b], int(tl[b]), int(ul[b]))
        c.backward()
        assert costs[b] == pytest.approx(c.item(), rel=1e-9)
        g = grads[b]
        np.testing.assert_allclose(g[: tl[b], : ul[b] + 1, :V], x.grad[: tl[b], : ul[b] + 1].numpy(), atol=1e-6)
        # zero outside the valid lattice and in the padded columns
        assert np.all(g[tl[b]:] == 0) and np.all(g[:, ul[b] + 1:] == 0) and np.all(g[..., V:] == 0)
        # alpha(T-1,U)+lp_blank == beta(0,0)
        assert be[b, 0, 0] == pytest.approx(-costs[b], rel=1e-12)
        # every lattice row's gradient sums to ~0 (softmax-fused gradient)
        np.testing.assert_allclose(g[..., :V].sum(-1), 0, atol=1e-6)


def test_length_rounding_matches_reference_call_site():
    """speechbrain/nnet/losses.py:58-59: (rel*dim).round().int() on float32 tensors."""
    rel = np.array([1.0, 0.9, 0.8, 0.7, 0.5], np.float32)
    assert RR.abs_lengths(rel, 50).tolist() == (torch.from_numpy(rel) * 50).round().int().tolist()
    assert RR.abs_lengths(rel, 21).tolist() == (torch.from_numpy(rel) * 21).round().int().tolist()


def test_invalid_lengths_raise():
    lg = np.zeros((1, 2, 2, 3), np.float32)
    with pytest.raises(ValueError):
        RR.rnnt_costs_grads(lg, np.ones((1, 1), np.int32), [3], [1], 0)
