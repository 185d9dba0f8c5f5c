#!/usr/bin/env python3
"""FFN-shaped GEMMs (M = 8000 / 4000 rows, d_model 256, d_ffn 2048) on the 8-wave wide-tile kernel (csrc/gemm_big.hip) against the
4-wave tiles of csrc/gemm.hip (TSASR_GEMM_BIG=0 in a second process): graph-replayed, hot operands, GPU time per launch."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("ts-asr_amd.ops")
DEV = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for M in (8000, 4000):
    F1, D = 2048, 256
    x = torch.randn(M, D, device=DEV).to(torch.bfloat16); w1 = torch.randn(F1, D, device=DEV).to(torch.bfloat16)
    b1 = torch.randn(F1, device=DEV); h = torch.randn(M, F1, device=DEV).to(torch.bfloat16)
    do = torch.randn(M, D, device=DEV).to(torch.bfloat16); w2t = torch.randn(F1, D, device=DEV).to(torch.bfloat16)
    dbias = torch.zeros(F1, device=DEV)
    fl = 2.0 * M * F1 * D
    for name, fn in [
        ("plain      ", lambda: ops.gemm_bf16(x, w1, M, F1, D, D, D, 0, 0)),
        ("fused<1> .1", lambda: ops.gemm_bf16_fused(x, w1, M, F1, D, D, D, 0, 0, 1, bias=b1, slope=0.01, p=0.1, seed=5)),
        ("fused<2> .1", lambda: ops.gemm_bf16_fused(do, w2t, M, F1, D, D, D, 0, 0, 2, y=h, slope=0.01, p=0.1, seed=5, dbias=dbias)),
        ("down K=2048", lambda: ops.gemm_bf16(h, w2t.view(D, F1), M, D, F1, F1, F1, 0, 0)),
    ]:
        us = timeit(fn)
        print(f"M={M} {name}: {us:6.1f} us  {fl / us / 1e6:6.0f} TFLOP/s")
