"""Python face of the RNN-T loss oracle (TEST INFRASTRUCTURE; see rnnt_ref.c for the algorithm notes).

* ``rnnt_costs_grads``   - ctypes call into oracle/_build/librnnt_ref.so (compiled from rnnt_ref.c by gcc)
* ``transducer_loss_ref``- the reference's call-site semantics, speechbrain/nnet/losses.py:58-79:
                           abs lengths = round(rel * dim), blank index, reduction "mean" over the batch
* ``RnntLossRefFn``      - autograd wrapper so the CPU baseline step can back-propagate through it
* ``brute_force_cost``   - float64 enumeration of every alignment path (tiny lattices only)
"""
import ctypes
import itertools
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librnnt_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "rnnt_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.rnnt_ref_loss.restype = ctypes.c_int
    return _lib


def rnnt_costs_grads(logits, targets, tlen, ulen, blank=0, V=None, want_grads=True, want_ab=False):
    """logits [B,T,U1,ldl] float32 (first V of each row used); returns (costs f64 [B], grads f32 or None[, alpha, beta])."""
    lg = np.ascontiguousarray(logits, np.float32)
    B, T, U1, ldl = lg.shape
    V = ldl if V is None else V
    tg = np.ascontiguousarray(targets, np.int32).reshape(B, -1)
    tl = np.ascontiguousarray(tlen, np.int32)
    ul = np.ascontiguousarray(ulen, np.int32)
    costs = np.zeros(B, np.float64)
    grads = np.zeros_like(lg) if want_grads else None
    al = np.zeros((B, T, U1), np.float64) if want_ab else None
    be = np.zeros((B, T, U1), np.float64) if want_ab else None
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rc = _load().rnnt_ref_loss(p(lg), B, T, U1, V, ldl, p(tg), tg.shape[1], p(tl), p(ul), int(blank),
                               p(costs), p(grads), p(al), p(be))
    if rc != 0:
        raise ValueError(f"rnnt_ref_loss: invalid lengths (rc={rc})")
    return (costs, grads, al, be) if want_ab else (costs, grads)


def abs_lengths(rel, dim):
    """(rel * dim).round().int()  - speechbrain/nnet/losses.py:58-59 (torch.round = half-to-even, on float32)."""
    return np.rint(np.asarray(rel, np.float32) * np.float32(dim)).astype(np.int32)


def transducer_loss_ref(logits, targets, input_lens, target_lens, blank_index=0, reduction="mean"):
    lg = np.asarray(logits, np.float32)
    tg = np.asarray(targets)
    tl = abs_lengths(input_lens, lg.shape[1])
    ul = abs_lengths(target_lens, tg.shape[1])
    costs, grads = rnnt_costs_grads(lg, tg, tl, ul, blank_index)
    if reduction == "mean":
        return costs.mean(), grads / lg.shape[0]
    if reduction == "sum":
        return costs.sum(), grads
    return costs, grads


def brute_force_cost(logits, targets, T, U, blank=0):
    """-log sum over all monotone alignments, float64, exponential time: for lattices up to ~5x4."""
    lg = np.asarray(logits, np.float64)
    lp = lg - np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1, keepdims=True)) - lg.max(-1, keepdims=True)
    total = -np.inf
    # a path = positions of the U emissions among T+U moves; it must end with the final blank at (T-1,U)
    for emits in itertools.combinations(range(T + U - 1), U):
        t = u = 0
        s = 0.0
        es = set(emits)
        for step in range(T + U - 1):
            if step in es:
                s += lp[t, u, targets[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
        if t != T - 1 or u != U:
            continue
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return -total


try:  # torch is only needed for the autograd face
    import torch

    class RnntLossRefFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, logits, targets, tlen, ulen, blank):
            costs, grads = rnnt_costs_grads(logits.detach().cpu().numpy(), targets.cpu().numpy(),
                                            tlen.cpu().numpy(), ulen.cpu().numpy(), blank)
            ctx.save_for_backward(torch.from_numpy(grads))
            return torch.from_numpy(costs).to(torch.float32)

        @staticmethod
        def backward(ctx, gcosts):
            (g,) = ctx.saved_tensors
            return g * gcosts.view(-1, 1, 1, 1), None, None, None, None

    def transducer_loss_ref_torch(logits, targets, input_lens, target_lens, blank_index=0, reduction="mean"):
        tl = (input_lens * logits.shape[1]).round().int()
        ul = (target_lens * targets.shape[1]).round().int()
        costs = RnntLossRefFn.apply(logits, targets.int(), tl, ul, blank_index)
        return costs.mean() if reduction == "mean" else costs.sum() if reduction == "sum" else costs
except ImportError:  # pragma: no cover
    pass
