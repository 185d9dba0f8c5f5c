"""Forced macro-tile sweep at the long-K / narrow-N forward shapes (tsasr_gemm_set_plan)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("ts-asr_amd.ops"); C = importlib.import_module("ts-asr_amd._capi")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def timeit(fn, n=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best
for (M, N, K) in [(8000, 256, 2048), (8000, 256, 256), (4000, 256, 2048), (4000, 256, 256), (8000, 768, 256), (8000, 512, 256), (4000, 768, 256), (8000, 256, 768)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm_bf16(A, B, M, N, K, K, K, 0, 0, out=out)
    res = []
    for tile in (-1, 0, 1, 2):
        C.lib().tsasr_gemm_set_plan(tile, 0)
        res.append(timeit(fn))
    C.lib().tsasr_gemm_set_plan(-1, 0)
    print(f"M={M} N={N} K={K}: auto {res[0]:.1f}  128x128 {res[1]:.1f}  128x64 {res[2]:.1f}  64x64 {res[3]:.1f} us", flush=True)
