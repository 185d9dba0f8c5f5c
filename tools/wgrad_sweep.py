#!/usr/bin/env python3
"""Sweep macro-tile x split-K x main loop for the weight-gradient GEMMs (dW = dY^T . X, fp32 accumulate) at the model's shapes."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops"); C = importlib.import_module("ts-asr_amd._capi")
from tools.gemm_bench import timeit  # noqa
DEV = "cuda"
SHAPES = [(2048, 256, 8000), (256, 2048, 8000), (768, 256, 8000), (512, 256, 8000), (256, 256, 8000), (256, 512, 8000)]
for (M, N, K) in SHAPES:
    A = torch.randn(K, M, device=DEV).to(torch.bfloat16)
    B = torch.randn(K, N, device=DEV).to(torch.bfloat16)
    out = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    fn = lambda: ops.gemm_bf16(A, B, M, N, K, M, N, 1, 1, out=out, accumulate=True)
    C.lib().tsasr_gemm_set_plan(-1, 0); C.lib().tsasr_gemm_set_ring(1)
    res = [("auto", timeit(fn))]
    for tile in (0, 1, 2):
        for splits in (2, 4, 8, 16, 31):
            for ring in (0, 2):
                C.lib().tsasr_gemm_set_plan(tile, splits); C.lib().tsasr_gemm_set_ring(ring)
                try:
                    res.append((f"t{tile} s{splits:2d} {'ring' if ring else 'reg '}", timeit(fn)))
                except Exception as e:   # workspace sized by plan(): consistent, but keep the sweep going
                    res.append((f"t{tile} s{splits} r{ring} ERR {type(e).__name__}", 1e9))
    C.lib().tsasr_gemm_set_plan(-1, 0); C.lib().tsasr_gemm_set_ring(1)
    res.sort(key=lambda x: x[1])
    print(f"dW[{M}x{N}] K={K}: auto {dict(res)['auto']:.1f} us | best: " + ", ".join(f"{n} {t:.1f}" for n, t in res[:6]))
