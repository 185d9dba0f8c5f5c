#!/usr/bin/env python3
"""Cross-graph dependency through EXTERNAL event nodes (event record / wait captured as graph nodes): graph A = chain, record E, chain;
graph B = chain, wait E, chain; launched on two streams without any host-side event call in between."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
ops = importlib.import_module(entry.PKG + ".ops")
dev = "cuda:0"
w = (torch.randn(256, 256, device=dev) * 0.05).to(torch.bfloat16)
g_, b_ = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def chain(x, n):
    for _ in range(n):
        x = ops.layer_norm(ops.matmul_nt(x, w), g_, b_, 1e-5)
    return x


xs = [torch.randn(4000, 256, device=dev).to(torch.bfloat16) for _ in range(2)]
flag = torch.zeros(1, device=dev)
out = torch.zeros(1, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
E = torch.cuda.Event(external=True)
with torch.no_grad():
    for x in xs:
        chain(x, 4)
    torch.cuda.synchronize()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=streams[0]):
        chain(xs[0], 30)
        flag.add_(1.0)
        E.record(streams[0])
        chain(xs[0], 30)
    with torch.cuda.graph(gb, stream=streams[1]):
        chain(xs[1], 30)
        streams[1].wait_event(E)
        out.copy_(flag)
        chain(xs[1], 30)
torch.cuda.synchronize()
print("captured")


def run(reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(streams[0]):
            ga.replay()
        with torch.cuda.stream(streams[1]):
            gb.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


run(3)
flag.zero_(); torch.cuda.synchronize()
t = run(20)
print(f"two graphs with an external event between them: {t:.3f} ms per pair; flag {float(flag)} out {float(out)} (out must equal flag: B's second half saw A's first half)")
