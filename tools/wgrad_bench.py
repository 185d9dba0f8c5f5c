#!/usr/bin/env python3
"""Time the grouped weight-gradient launch (csrc/wgrad.hip) on the job mix of one training step at BASELINE configs[1] against the
per-weight split-K path of round 1 (tsasr_gemm_bf16 transA=transB=1 + slab reduction). Interleaved rounds in one process."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("ts-asr_amd.ops")
DEV = torch.device("cuda:0")


def layer_jobs(tokens):
    return [(2048, 256, tokens), (256, 2048, tokens)] * 2 + [(768, 256, tokens), (256, 256, tokens), (512, 256, tokens), (256, 256, tokens)]


def main():
    ops.reduce_defer_prepare(DEV)
    shapes = layer_jobs(8000) * int(os.environ.get("LAYERS", 3)) + layer_jobs(4000) * int(os.environ.get("SLAYERS", 0))
    g = torch.Generator().manual_seed(0)
    jobs = []
    shared = os.environ.get("SHARED", "0") != "0"   # every job reads the same two buffers: operands stay cache-resident (compute-side rate)
    pool = {}
    for (M, N, K) in shapes:
        if shared:
            dy = pool.setdefault(("dy", K, M), torch.randn(K, M, generator=g).to(torch.bfloat16).to(DEV))
            x = pool.setdefault(("x", K, N), torch.randn(K, N, generator=g).to(torch.bfloat16).to(DEV))
        else:
            dy, x = torch.randn(K, M, generator=g).to(torch.bfloat16).to(DEV), torch.randn(K, N, generator=g).to(torch.bfloat16).to(DEV)
        jobs.append((dy, x, torch.zeros(M, N, device=DEV), torch.nn.Parameter(torch.empty(0))))
    flops = sum(2.0 * M * N * K for M, N, K in shapes)

    def grouped(e0=None):
        for dy, x, w, p in jobs:
            ops.wgrad_queue(p, w, dy, x)
        if e0 is not None:
            e0.record()       # GPU time of the launch only: the queueing above is host work that overlaps earlier kernels in a real step
        ops.wgrad_flush()

    def per_weight():
        for dy, x, w, p in jobs:
            M, N = w.shape
            ops.gemm_bf16(dy, x, M, N, dy.shape[0], M, N, 1, 1, out=w, accumulate=True)

    res = {"grouped": [], "per_weight": []}
    for fn in (grouped, per_weight):
        fn()
    torch.cuda.synchronize()
    only = os.environ.get("ONLY")
    for _ in range(int(os.environ.get("ROUNDS", 8))):
        for name, fn in (("grouped", grouped), ("per_weight", per_weight)):
            if only and name != only:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if name == "grouped":
                fn(e0)
            else:
                e0.record()
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1))
    for name, ms in res.items():
        if not ms:
            continue
        ms.sort()
        print(f"{name:12s} {len(jobs)} GEMMs {flops / 1e9:.1f} GFLOP: median {ms[len(ms) // 2] * 1e3:.1f} us, min {ms[0] * 1e3:.1f} us -> {flops / ms[len(ms) // 2] / 1e9:.0f} TFLOP/s")


    if os.environ.get("TSASR_WGRAD_DEBUG", "0") != "0":
        import ctypes
        from importlib import import_module
        C = import_module("ts-asr_amd._capi")
        L = ctypes.CDLL(C.LIB_PATH)
        buf = (ctypes.c_ulonglong * 4)()
        torch.cuda.synchronize()
        L.tsasr_wgrad_debug_read(buf)
        if buf[1]:
            print(f"  wg0 main loop: {buf[0]} cycles, {buf[1] * 10} ns -> {buf[0] / (buf[1] * 10):.3f} GHz, {buf[0] / max(buf[2], 1):.0f} cycles / k-tile ({buf[2]} k-tiles)")


if __name__ == "__main__":
    main()
