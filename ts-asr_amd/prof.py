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


def event_overhead_ms(pairs=64):
    """Median elapsed time of an EMPTY event pair on the current stream: what a start/stop pair adds around a launch (the two
    timestamp packets and the dispatch gap between them, ~3 us) - subtracted from every region by collect()."""
    evs = []
    for _ in range(pairs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def collect(subtract_overhead=True):
    """{name: (count, mean_ms)} - call after torch.cuda.synchronize(). Durations are net of the empty-pair overhead."""
    out = {}
    ov = event_overhead_ms() if (subtract_overhead and _records) else 0.0
    for name, evs in _records.items():
        ms = [max(a.elapsed_time(b) - ov, 0.0) for a, b in evs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    out_overhead[0] = ov
    return out


out_overhead = [0.0]


def reset():
    _records.clear()
    _work.clear()
