#!/usr/bin/env python3
"""Per-k-tile cost of the GEMM main loops: one workgroup alone (latency) vs a full grid (throughput)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops"); C = importlib.import_module("ts-asr_amd._capi")
from tools.gemm_bench import timeit
DEV = "cuda"
for (M, N, K, ta, tb) in [(128, 128, 8192, 0, 0), (128, 128, 8192, 1, 1), (2048, 2048, 8192, 0, 0), (2048, 2048, 8192, 1, 1), (4096, 4096, 4096, 0, 0), (8192, 8192, 8192, 0, 0)]:
    A = torch.randn((K, M) if ta else (M, K), device=DEV).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=DEV).to(torch.bfloat16)
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    fn = lambda: ops.gemm_bf16(A, B, M, N, K, M if ta else K, N if tb else K, ta, tb, out=out)
    lib = lambda: torch.matmul(A.t() if ta else A, B if tb else B.t())
    C.lib().tsasr_gemm_set_plan(0, 1)
    r = {}
    for name, ring in (("ring", 2), ("reg", 0)):
        C.lib().tsasr_gemm_set_ring(ring); r[name] = timeit(fn, 5)
    C.lib().tsasr_gemm_set_plan(-1, 0); C.lib().tsasr_gemm_set_ring(1)
    r["lib"] = timeit(lib, 5)
    nk = K // 64
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} tA={ta} tB={tb}: " + " | ".join(f"{k} {v:8.1f} us ({v/nk*1e3:6.0f} ns/k-tile, {fl/v/1e6:6.0f} TF)" for k, v in r.items()))
