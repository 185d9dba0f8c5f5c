#!/usr/bin/env python3
"""Macro-tile x main-loop sweep for the forward / input-gradient GEMM shapes (both operands k-contiguous), plain and fused epilogue."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops"); C = importlib.import_module("ts-asr_amd._capi")
from tools.gemm_bench import timeit  # noqa (runs the standard table first)
DEV = "cuda"
for (M, N, K) in [(8000, 2048, 256), (8000, 256, 2048), (8000, 768, 256), (8000, 256, 256), (8000, 512, 256), (4000, 2048, 256), (4000, 256, 2048)]:
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16); B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    plain = lambda: ops.gemm_bf16(A, B, M, N, K, K, K, 0, 0, out=out)
    fused = lambda: ops.gemm_bf16_fused(A, B, M, N, K, K, K, 0, 0, 1, bias=bias, slope=0.01, p=0.1, seed=5)
    res = []
    for tile in (-1, 0, 1, 2):
        for ring in (0, 2) if tile >= 0 else (1,):
            C.lib().tsasr_gemm_set_plan(tile, 0); C.lib().tsasr_gemm_set_ring(ring)
            res.append((f"t{tile} {'auto' if tile < 0 else ('ring' if ring else 'reg ')}", timeit(plain), timeit(fused)))
    C.lib().tsasr_gemm_set_plan(-1, 0); C.lib().tsasr_gemm_set_ring(1)
    print(f"[{M}x{N}] K={K}: " + " | ".join(f"{n} {p:.1f}/{f:.1f}" for n, p, f in res))
