"""Opt-in per-kernel timing with HIP events on the stream the kernels are launched on (torch's current stream).
bench.py switches it on for the timed region to obtain the live average launch duration of a named kernel family
(the `roofline` object of its JSON line); off by default - zero overhead in training."""
import contextlib
import os

import torch

ENABLED = False
_records = {}
_work = {}
_bytes = {}


@contextlib.contextmanager
def region(name, work=0.0, nbytes=0.0):
    """``work`` = algorithmic FLOPs of this launch, ``nbytes`` = its algorithmic HBM bytes (operands read once, results written once);
    both are summed per name so that achieved rate = work / time."""
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
        _bytes[name] = _bytes.get(name, 0.0) + float(nbytes)


def work():
    return dict(_work)


def algorithmic_bytes():
    return dict(_bytes)


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


def collect(subtract_overhead=True, repeats=1, fastest=True):
    """{name: (count, mean_ms)} - call after torch.cuda.synchronize(). Durations are net of the empty-pair overhead. ``repeats``:
    the instrumented region ran the same launch sequence that many times (steps); each launch is then represented by its FASTEST
    repeat (a launch that happened to queue behind another stream's kernel otherwise drags the family's mean: one 3 ms stall among 120
    launches doubled it)."""
    out = {}
    # 0.8: with a kernel between the two records its own dispatch hides part of the gap an empty pair shows; calibrated against the
    # rocprofv3 durations of the same launches (wave-K weight gradient 19.5 us, 128x64 ring GEMM 12.4 us, attention forward 28.7 us:
    # net event times 19.4 / 12.7 / 30.5 us) - errs on the slow side
    ov = 0.8 * event_overhead_ms() if (subtract_overhead and _records) else 0.0
    for name, evs in _records.items():
        ms = [max(a.elapsed_time(b) - ov, 0.0) for a, b in evs]
        if fastest and repeats > 1 and len(ms) % repeats == 0:
            n = len(ms) // repeats
            best = [min(ms[i + r * n] for r in range(repeats)) for i in range(n)]
            out[name] = (len(ms), sum(best) / max(n, 1))
        else:
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    out_overhead[0] = ov
    return out


out_overhead = [0.0]


def reset():
    _records.clear()
    _work.clear()
    _bytes.clear()


# ---- phase stamps inside a captured step (tools/step_stamps.py) ------------------------------------------------------------------------
# TSASR_STAMPS=1: the recipe drops a one-thread kernel that stores the device wall clock at a few points of the step, on whatever stream
# is current there. Replayed with the graph, they give the true start time of each phase without a profiler attached.
STAMPS = os.environ.get("TSASR_STAMPS", "0") == "1"
_stamp_buf = None
_stamp_names = []


def stamp_begin(device):
    """Start of a step: (re)use a 256-slot buffer (allocated once, before any capture)."""
    global _stamp_buf
    if not STAMPS:
        return
    if _stamp_buf is None:
        _stamp_buf = torch.zeros(256, dtype=torch.int64, device=device)
    _stamp_names.clear()


def stamp(name):
    if not STAMPS or _stamp_buf is None or len(_stamp_names) >= 256:
        return
    from . import _capi as C
    k = len(_stamp_names)
    _stamp_names.append(name)
    if C.lab().tsasr_lab_stamp(C.ptr(_stamp_buf[k:k + 1]), C.stream_ptr()) != 0:      # lab equipment (include/tsasr_lab.h), tools only
        raise C.TsasrHipError("tsasr_lab_stamp failed")


def stamps_us():
    """[(name, microseconds since the first stamp)] of the last step that ran (host read)."""
    if _stamp_buf is None or not _stamp_names:
        return []
    t = _stamp_buf[:len(_stamp_names)].cpu().tolist()
    return [(n, (v - t[0]) / 100.0) for n, v in zip(_stamp_names, t)]
