#!/usr/bin/env python3
"""A few launches of the model's GEMM shapes for rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE per kernel)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops")
DEV = "cuda"
def run(M, N, K, ta, tb, f32, n=5):
    A = torch.randn((K, M) if ta else (M, K), device=DEV).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=DEV).to(torch.bfloat16)
    out = torch.zeros(M, N, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
    for _ in range(n):
        ops.gemm_bf16(A, B, M, N, K, M if ta else K, N if tb else K, ta, tb, out=out, accumulate=bool(f32))
    torch.cuda.synchronize()
run(2048, 256, 8000, 1, 1, 1)     # FFN weight gradient
run(8000, 2048, 256, 0, 0, 0)     # FFN first projection
run(8000, 256, 2048, 0, 0, 0)     # FFN second projection
run(8000, 256, 2048, 0, 1, 0)     # its dgrad
