"""Do two independent kernel chains captured on two streams overlap on replay? (order of capture: sequential vs interleaved)"""
import time, torch
dev = "cuda:0"
N = 60
def work(x):   # a kernel that under-fills the GPU: 64 workgroups
    return torch.nn.functional.layer_norm(x, (1024,))
xa = torch.randn(64 * 4, 1024, device=dev); xb = torch.randn(64 * 4, 1024, device=dev)
big_a = torch.randn(8000, 2048, device=dev); big_b = torch.randn(8000, 2048, device=dev)
def chain(x, n):
    for _ in range(n):
        x = torch.nn.functional.layer_norm(x, (x.shape[-1],))
    return x
side = torch.cuda.Stream()
def build(mode, xa, xb):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        xa.add_(0.0)                           # root
        side.wait_stream(cur)
        if mode == "single":
            ya = chain(xa, N); yb = chain(xb, N)
        elif mode == "seq":                    # all of B (side) captured first, then all of A (main)
            with torch.cuda.stream(side):
                yb = chain(xb, N)
            ya = chain(xa, N)
        elif mode == "interleaved":
            ya, yb = xa, xb
            for _ in range(N):
                with torch.cuda.stream(side):
                    yb = chain(yb, 1)
                ya = chain(ya, 1)
        if mode != "single":
            cur.wait_stream(side)
        out = ya.sum() + yb.sum()
    return g
for label, a, b in (("small kernels (256 rows)", xa, xb), ("GPU-filling kernels (8000x2048)", big_a, big_b)):
    for mode in ("single", "seq", "interleaved"):
        chain(a, 2); chain(b, 2)
        with torch.cuda.stream(side):
            chain(b, 2)
        torch.cuda.synchronize()
        g = build(mode, a, b)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print(f"{label:34s} {mode:12s} {(time.perf_counter() - t0) / 5 * 1e3:7.3f} ms per replay", flush=True)
