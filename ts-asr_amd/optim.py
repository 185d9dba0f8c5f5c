"""Optimizer step over the flat arena: global-norm clipping + AdamW in one pass.

Reference: speechbrain/core.py:1082-1093 (torch.nn.utils.clip_grad_norm_(max_grad_norm) -> optimizer.step() ->
zero_grad) with opt_class = torch.optim.AdamW(lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
(hparams conformer-t_scratch.yaml:267-271); >= 3 passes over 51 M parameters x (p, g, m, v) and a host sync for
grad_norm.item(). Here the norm stays on the device and clip + decoupled weight decay + Adam moments + update are one
kernel over four flat buffers (``tsasr_clip_adamw_step`` in csrc/optim.hip).
"""
import functools

import torch

from . import _capi as C
from . import prof


_HYPER_SLOTS = 64

class FusedClipAdamW:
    """Exposes ``param_groups`` (the Noam scheduler writes ``lr`` there) and ``step()``; state = flat m, v."""

    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0, skip_nonfinite=False):
        self.arena = arena
        # False (default) = the reference: a step whose gradient norm is not finite is applied like any other (SB/core.py:1072-1093);
        # True: the kernel leaves parameters and moments alone for such a step and counts it (take_skipped_steps)
        self.skip_nonfinite = bool(skip_nonfinite)
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "params": arena.params_ordered}]
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.exp_avg = torch.zeros_like(arena.grads)
        self.exp_avg_sq = torch.zeros_like(arena.grads)
        arena.companions += [self.exp_avg, self.exp_avg_sq]  # follow the arena's one-time re-layout
        self.t = 0
        self._norm_buf = torch.zeros(2, device=arena.device)        # [grad norm, steps skipped for a non-finite norm]: device memory, never synced in the step
        self.last_grad_norm = self._norm_buf[0]
        self._hyper_host = self._hyper_dev = self._ws = None

    def prepare(self):
        """Host half of a step: advance t and ship {lr, 1-b1^t, 1-b2^t} to the device (pinned staging -> async copy). Kept
        outside any captured hipGraph so that the Noam schedule keeps acting on replays."""
        g = self.param_groups[0]
        self.t += 1
        b1, b2 = g["betas"]
        cuda = self.arena.device.type == "cuda"
        if self._hyper_host is None:
            # a RING of pinned staging rows: with graph replay the host runs many steps ahead of the GPU, and an asynchronous copy reads
            # its pinned source when it EXECUTES - one reused row handed step i the learning rate of step i+k (run-to-run different
            # losses in un-synchronised loops). A row is rewritten only after the copy that last used it has completed.
            self._hyper_host = torch.empty(_HYPER_SLOTS, 3, dtype=torch.float32).pin_memory() if cuda else torch.empty(_HYPER_SLOTS, 3)
            self._hyper_events = [None] * _HYPER_SLOTS
            self._hyper_dev = torch.empty(3, dtype=torch.float32, device=self.arena.device)
            self._ws = torch.empty(max(256, C.lib().tsasr_clip_adamw_workspace_bytes()), dtype=torch.uint8, device=self.arena.device)
        k = self.t % _HYPER_SLOTS
        if cuda and self._hyper_events[k] is not None:
            self._hyper_events[k].synchronize()
        row = self._hyper_host[k]
        row[0], row[1], row[2] = g["lr"], 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        self._hyper_dev.copy_(row, non_blocking=True)
        if cuda:
            ev = self._hyper_events[k] or torch.cuda.Event()
            ev.record()
            self._hyper_events[k] = ev

    def launch(self):
        """Device half: two kernel launches over the arena (graph-capturable)."""
        g = self.param_groups[0]
        a = self.arena
        b1, b2 = g["betas"]
        C.require_gpu(a.flat_params)
        with prof.region("clip_adamw"):
            C.check(C.lib().tsasr_clip_adamw_step(C.ptr(a.flat_params), C.ptr(a.flat_params16), C.ptr(a.grads), C.ptr(self.exp_avg),
                                                  C.ptr(self.exp_avg_sq), C.ptr(self._hyper_dev), C.ptr(self._norm_buf[0:1]),
                                                  C.ptr(self._norm_buf[1:2]) if self.skip_nonfinite else None, a.numel,
                                                  float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), self.max_grad_norm,
                                                  C.ptr(self._ws), self._ws.numel(), C.stream_ptr()), "tsasr_clip_adamw_step")
            a.refresh_transposed()

    def step(self):
        self.prepare()
        self.launch()

    def take_skipped_steps(self):
        """Steps the kernel skipped because the gradient norm was not finite, since the last call (one host read)."""
        n = int(self._norm_buf[1].item())
        if n:
            self._norm_buf[1].zero_()
        return n

    def zero_grad(self, set_to_none=False):
        self.arena.zero_()

    def state_dict(self):
        return {"t": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "param_groups": [
            {k: v for k, v in self.param_groups[0].items() if k != "params"}]}


def _clip_adamw(p, g, m, v, norm_out, lr, b1, b2, eps, wd, t, max_norm, p16=None, skipped_out=None):
    """Stand-alone call of csrc/optim.hip on explicit buffers (unit tests). norm_out / skipped_out: one device float each (or None)."""
    C.require_gpu(p, g, m, v)
    hyper = torch.tensor([lr, 1.0 - b1 ** t, 1.0 - b2 ** t], dtype=torch.float32).to(p.device)
    ws = torch.empty(max(256, C.lib().tsasr_clip_adamw_workspace_bytes()), dtype=torch.uint8, device=p.device)
    C.check(C.lib().tsasr_clip_adamw_step(C.ptr(p), C.ptr(p16), C.ptr(g), C.ptr(m), C.ptr(v), C.ptr(hyper), C.ptr(norm_out), C.ptr(skipped_out), p.numel(),
                                          float(b1), float(b2), float(eps), float(wd), float(max_norm), C.ptr(ws), ws.numel(),
                                          C.stream_ptr()), "tsasr_clip_adamw_step")


class _WrappedTorchOptimizer:
    """Any other opt_class: used as given, clipping by torch (still one flat gradient buffer underneath)."""

    def __init__(self, opt, arena, max_grad_norm):
        self.opt, self.arena, self.max_grad_norm = opt, arena, float(max_grad_norm or 0.0)
        self.param_groups = opt.param_groups
        self.last_grad_norm = torch.zeros((), device=arena.device)

    def prepare(self):
        pass

    def launch(self):
        norm = self.arena.grad_norm()
        self.last_grad_norm.copy_(norm)
        if self.max_grad_norm > 0:
            self.arena.grads.mul_(torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0))
        self.opt.step()
        self.arena.sync_shadow()

    def step(self):
        self.launch()

    def zero_grad(self, set_to_none=False):
        self.arena.zero_()


def make_optimizer(opt_class, params, arena, max_grad_norm, skip_nonfinite=False):
    kw = opt_class.keywords if isinstance(opt_class, functools.partial) else {}
    base = opt_class.func if isinstance(opt_class, functools.partial) else opt_class
    if base is torch.optim.AdamW and not (set(kw) - {"lr", "betas", "eps", "weight_decay"}):
        return FusedClipAdamW(arena, max_grad_norm=max_grad_norm, skip_nonfinite=skip_nonfinite, **kw)
    return _WrappedTorchOptimizer(opt_class(params), arena, max_grad_norm)
