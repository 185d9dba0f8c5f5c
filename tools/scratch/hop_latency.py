"""Latency of a cross-stream dependency inside a replayed hipGraph: N tiny kernels in a chain on ONE stream vs alternating between TWO streams (scratch)."""
import time, torch
dev = "cuda:0"
x = torch.zeros(64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 50
def chain(alternate):
    cur = torch.cuda.current_stream()
    last = cur
    for i in range(N):
        st = s2 if (alternate and (i & 1)) else cur
        if st is not last:
            st.wait_stream(last)
        with torch.cuda.stream(st):
            x.add_(1.0)
        last = st
    if last is not cur:
        cur.wait_stream(last)
for alt in (False, True):
    g = torch.cuda.CUDAGraph()
    chain(alt); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        chain(alt)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"{'two streams, alternating' if alt else 'one stream':26s}: {dt * 1e6 / N:7.2f} us per kernel in the chain ({dt * 1e3:.2f} ms per replay of {N})")
