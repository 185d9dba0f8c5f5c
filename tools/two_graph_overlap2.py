#!/usr/bin/env python3
"""Which ingredient of core.GraphSegments serialises two independent graphs? (A) separate pools, no events; (B) shared pool;
(C) a 'start' graph + events + waits as GraphSegments.replay() issues them; (D) B + C."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
ops = importlib.import_module(entry.PKG + ".ops")
dev = "cuda:0"
n = 60
w = (torch.randn(256, 256, device=dev) * 0.05).to(torch.bfloat16)
g_, b_ = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def chain(x):
    for _ in range(n):
        x = ops.layer_norm(ops.matmul_nt(x, w), g_, b_, 1e-5)
    return x


xs = [torch.randn(4000, 256, device=dev).to(torch.bfloat16) for _ in range(2)]
tiny = torch.zeros(64, device=dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventDisableSystemFence = 0x2, 0x20000000


def raw_event(flags):
    ev = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(flags)) == 0
    return ev


def raw_record(ev, st):
    assert hip.hipEventRecord(ev, ctypes.c_void_p(st.cuda_stream)) == 0


def raw_wait(st, ev):
    assert hip.hipStreamWaitEvent(ctypes.c_void_p(st.cuda_stream), ev, ctypes.c_uint(0)) == 0


raw = {name: [raw_event(fl) for _ in range(3)] for name, fl in (("raw_fence", hipEventDisableTiming), ("raw_nofence", hipEventDisableTiming | hipEventDisableSystemFence))}
for shared in (False, True):
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    with torch.no_grad():
        for x in xs:
            chain(x)
        torch.cuda.synchronize()
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=streams[0]):
            tiny.add_(1.0)
        pool = g0.pool() if shared else None
        graphs = []
        for x, st in zip(xs, streams[1:]):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, pool=pool, stream=st):
                chain(x)
            graphs.append(gr)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event() for _ in range(3)]

    def run(mode, reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if mode == "events":
                with torch.cuda.stream(streams[0]):
                    g0.replay(); evs[0].record(streams[0])
                for i in (0, 1):
                    streams[1 + i].wait_event(evs[0])
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay(); evs[1 + i].record(streams[1 + i])
                streams[0].wait_event(evs[1]); streams[0].wait_event(evs[2])
            elif mode == "fork_only":      # start graph, event, both branches wait for it; no join
                with torch.cuda.stream(streams[0]):
                    g0.replay(); evs[0].record(streams[0])
                for i in (0, 1):
                    streams[1 + i].wait_event(evs[0])
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay()
            elif mode == "join_only":      # both branches, events, a joining graph on the third stream
                for i in (0, 1):
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay(); evs[1 + i].record(streams[1 + i])
                streams[0].wait_event(evs[1]); streams[0].wait_event(evs[2])
                with torch.cuda.stream(streams[0]):
                    g0.replay()
            elif mode == "late_record":    # both branches launched first, THEN the two event records and the join
                for i in (0, 1):
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay()
                for i in (0, 1):
                    evs[1 + i].record(streams[1 + i])
                streams[0].wait_event(evs[1]); streams[0].wait_event(evs[2])
                with torch.cuda.stream(streams[0]):
                    g0.replay()
            elif mode in raw:              # join only, events through the HIP API directly (with / without the system-scope fence)
                e = raw[mode]
                for i in (0, 1):
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay()
                    raw_record(e[1 + i], streams[1 + i])
                raw_wait(streams[0], e[1]); raw_wait(streams[0], e[2])
                with torch.cuda.stream(streams[0]):
                    g0.replay()
            elif mode == "plain":
                for i in (0, 1):
                    with torch.cuda.stream(streams[1 + i]):
                        graphs[i].replay()
            else:
                with torch.cuda.stream(streams[1]):
                    graphs[0].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    run("plain"); run("events")
    print(f"shared pool={shared}: one chain {run('one'):.3f} ms; two graphs plain {run('plain'):.3f} ms; with start graph + events {run('events'):.3f} ms; fork only {run('fork_only'):.3f}; join only {run('join_only'):.3f}; late record {run('late_record'):.3f}; raw events {run('raw_fence'):.3f}; raw events without system fence {run('raw_nofence'):.3f}")
