"""Opt-in per-kernel timing with HIP events on the stream the kernels are launched on (torch's current stream).
bench.py switches it on for the timed region to obtain the live average launch duration of a named kernel family
(the `roofline` object of its JSON line); off by default - zero overhead in training."""
import contextlib

import torch

ENABLED = False
_records = {}
_work = {}


@contextlib.contextmanager
def region(name, work=0.0):
    """``work`` = algorithmic FLOPs (or bytes) of this launch; summed per name so that achieved rate = work / time."""
    if not ENABLED:
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _records.setdefault(name, []).append((a, b))
        _work[name] = _work.get(name, 0.0) + float(work)


def work():
    return dict(_work)


def collect():
    """{name: (count, mean_ms)} - call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in _records.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


def reset():
    _records.clear()
    _work.clear()
