#!/usr/bin/env python3
"""What each piece of the FFN GEMMs' fused epilogues costs (csrc/gemm_big.hip; M = 8000, N = 2048, K = 256, hot operands, 20 launches per
graph replay, best of 5): plain product, + bias + LeakyReLU, + dropout, + mask words; the data gradient plain / with mask words / from the
saved activation. At this shape a thread owns 256 outputs and a SIMD two waves: one VALU operation per element = 0.85 us."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops")
DEV = "cuda"


def timeit(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n * 1e3)
    return best


M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (8000, 2048, 256)
A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
B = (torch.randn(N, K, device=DEV) / 16).to(torch.bfloat16)
bias = torch.randn(N, device=DEV)
mask = torch.empty(M, N // 8, dtype=torch.int16, device=DEV)
print("plain mode 0             %6.1f us" % timeit(lambda: ops.gemm_bf16(A, B, M, N, K, K, K, 0, 0)))
print("bias + lrelu, p = 0      %6.1f us" % timeit(lambda: ops.gemm_bf16_fused(A, B, M, N, K, K, K, 0, 0, 1, bias=bias, slope=0.01, p=0.0, seed=5)))
print("bias + lrelu, p = 0.1    %6.1f us" % timeit(lambda: ops.gemm_bf16_fused(A, B, M, N, K, K, K, 0, 0, 1, bias=bias, slope=0.01, p=0.1, seed=5)))
print("... + mask words         %6.1f us" % timeit(lambda: ops.gemm_bf16_fused(A, B, M, N, K, K, K, 0, 0, 1, bias=bias, slope=0.01, p=0.1, seed=5, mask=mask)))
G = torch.randn(M, K, device=DEV).to(torch.bfloat16)
W2t = (torch.randn(N, K, device=DEV) / 16).to(torch.bfloat16)
Y = ops.gemm_bf16_fused(A, B, M, N, K, K, K, 0, 0, 1, bias=bias, slope=0.01, p=0.1, seed=5, mask=mask)
db = torch.empty(N, device=DEV)
print("dgrad plain              %6.1f us" % timeit(lambda: ops.gemm_bf16(G, W2t, M, N, K, K, K, 0, 0)))
print("dgrad mode 2 + mask      %6.1f us" % timeit(lambda: ops.gemm_bf16_fused(G, W2t, M, N, K, K, K, 0, 0, 2, y=Y, slope=0.01, p=0.1, seed=5, dbias=db, mask=mask)))
print("dgrad mode 2, no mask    %6.1f us" % timeit(lambda: ops.gemm_bf16_fused(G, W2t, M, N, K, K, K, 0, 0, 2, y=Y, slope=0.01, p=0.1, seed=5, dbias=db)))
