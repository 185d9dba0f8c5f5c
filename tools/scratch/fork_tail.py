"""Does ONE fork + join anywhere in a captured graph change what the kernels of its long single-branch tail cost? (scratch)"""
import time, torch
dev = "cuda:0"
import sys
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, y = torch.zeros(n, device=dev), torch.zeros(64, device=dev)
s2 = torch.cuda.Stream()
N = 400
def body(fork):
    cur = torch.cuda.current_stream()
    x.add_(1.0)
    if fork:
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            y.add_(1.0)
        x.add_(1.0)
        cur.wait_stream(s2)
    for _ in range(N):
        x.add_(1.0)
for fork in (False, True):
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
    print(f"{'one fork + join, then' if fork else 'no fork,':22s} a chain of {N}: {dt * 1e3:.3f} ms per replay = {dt * 1e6 / N:.2f} us per kernel")
