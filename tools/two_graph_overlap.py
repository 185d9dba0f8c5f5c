#!/usr/bin/env python3
"""Do SHORT kernels of two hipGraphs launched on two streams overlap? (Inside ONE captured graph ROCm 7.2 runs the short kernels of
forked branches one after the other - profiles/r02_notes.md section 5.) Chain = n x (GEMM [M,256]x[256,256] + LayerNorm) on the HIP ops."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
ops = importlib.import_module(entry.PKG + ".ops")
dev = "cuda:0"
M1, M2, n = int(sys.argv[1]), int(sys.argv[2]), 60
w = (torch.randn(256, 256, device=dev) * 0.05).to(torch.bfloat16)
g, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def chain(x):
    for _ in range(n):
        x = ops.layer_norm(ops.matmul_nt(x, w), g, b, 1e-5)
    return x


xs = [torch.randn(M, 256, device=dev).to(torch.bfloat16) for M in (M1, M2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
graphs = []
with torch.no_grad():
    for x, st in zip(xs, streams):
        with torch.cuda.stream(st):
            chain(x)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                y = chain(x)
            graphs.append(gr)
torch.cuda.synchronize()


def run(which, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in which:
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    a, b_, ab = run([0]), run([1]), run([0, 1])
print(f"chain of {2 * n} kernels: M={M1} alone {a:.3f} ms, M={M2} alone {b_:.3f} ms, both graphs on two streams {ab:.3f} ms (sum {a + b_:.3f})")
# same two chains captured as forked branches of ONE graph
with torch.no_grad():
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        cur = torch.cuda.current_stream()
        streams[1].wait_stream(cur)
        with torch.cuda.stream(streams[1]):
            chain(xs[1])
        chain(xs[0])
        cur.wait_stream(streams[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    gr.replay()
torch.cuda.synchronize()
print(f"one graph with the two chains as forked branches: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
