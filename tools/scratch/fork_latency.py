"""Cost of a fork + join inside a replayed hipGraph: N x [A -> (B || C) -> D] of tiny kernels against N x [A -> B -> C -> D] on one stream (scratch)."""
import time, torch
dev = "cuda:0"
x, y = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
s2 = torch.cuda.Stream()
N = 40
def body(fork):
    cur = torch.cuda.current_stream()
    for i in range(N):
        x.add_(1.0)                      # A
        if fork:
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                y.add_(1.0)              # C on the side
            for _ in range(NB): x.add_(1.0)      # B (NB kernels on the forking stream)
            cur.wait_stream(s2)
        else:
            for _ in range(NB): x.add_(1.0)
            y.add_(1.0)
        x.add_(1.0)                      # D
import itertools
for NB, fork in itertools.product((1, 8, 24), (False, True)):
    g = torch.cuda.CUDAGraph()
    body(fork); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        body(fork)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); R = 20
    for _ in range(R): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"B = {NB:2d} kernels, {'fork + join' if fork else 'one stream':12s}: {dt * 1e6 / N:7.2f} us per [A, B, C, D] group ({dt * 1e3:.3f} ms per replay)")
