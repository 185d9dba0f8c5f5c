#!/usr/bin/env python3
"""Micro-benchmark of csrc/gemm.hip at the model's shapes (interleaved rounds in one process, each timing = 20 launches captured into one
hipGraph). `lib` = torch.matmul (hipBLASLt / rocBLAS) on the same operands: a YARDSTICK only, never dispatched by the package.
  python tools/gemm_bench.py            the shapes of the step, default kernels vs the library
  python tools/gemm_bench.py --nn128    main-loop variants of the 128x64 tile (ring slots x wave-K split) and its lab floors (LDS-DMA ring
                                        alone / fragment reads + MFMA alone) at the N = 256 shapes of the step"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops")
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda"
SHAPES = [  # (M, N, K, tA, tB, out f32?)
    (8000, 2048, 256, 0, 0, 0), (8000, 256, 2048, 0, 0, 0), (8000, 768, 256, 0, 0, 0), (8000, 512, 256, 0, 0, 0),
    (8000, 256, 256, 0, 0, 0), (8000, 256, 2560, 0, 0, 0), (8000, 640, 256, 0, 0, 0),
    (8000, 256, 2048, 0, 1, 0), (8000, 2048, 256, 0, 1, 0), (8000, 256, 768, 0, 1, 0), (8000, 256, 256, 0, 1, 0),
    (4000, 256, 2048, 0, 0, 0), (4000, 2048, 256, 0, 0, 0), (4000, 256, 256, 0, 0, 0),
    (2048, 256, 8000, 1, 1, 1), (256, 2048, 8000, 1, 1, 1), (768, 256, 8000, 1, 1, 1), (256, 256, 8000, 1, 1, 1),
    (160000, 128, 1152, 0, 0, 0), (128, 1152, 160000, 1, 1, 1), (160000, 1152, 128, 0, 1, 0),
]


def timeit(fn, n=20):
    """Device time per call: n calls captured into one hipGraph (no host launch cost between them), best of 3 replays."""
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


def operands(M, N, K, ta, tb, f32):
    A = torch.randn((K, M) if ta else (M, K), device=DEV).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=DEV).to(torch.bfloat16)
    out = torch.zeros(M, N, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
    return A, B, out


if "--nn128" in sys.argv:
    variants = [(0, 0, 0), (4, 0, 0), (4, 2, 0), (3, 3, 0), (4, 3, 0), (5, 3, 0), (6, 3, 0), (4, 0, 1), (4, 0, 2), (4, 2, 2)]
    for (M, N, K) in [(8000, 256, 2048), (8000, 256, 256), (8000, 256, 768), (8000, 768, 256), (8000, 512, 256), (4000, 256, 2048)]:
        A, B, out = operands(M, N, K, 0, 0, 0)
        C.lib().tsasr_gemm_set_plan(1, 1)         # force the 128x64 tile for every shape of this sweep
        ref = (A.float() @ B.float().t())
        best = {v: 1e9 for v in variants}
        fn = lambda: ops.gemm_bf16(A, B, M, N, K, K, K, 0, 0, out=out)
        for rnd in range(3):
            for v in variants:
                C.lib().tsasr_gemm_set_nn128(*v)
                best[v] = min(best[v], timeit(fn))
        libt = min(timeit(lambda: torch.matmul(A, B.t())) for _ in range(2))
        errs = {}
        for v in variants:
            if v[2] == 0:
                C.lib().tsasr_gemm_set_nn128(*v); out.zero_(); fn(); torch.cuda.synchronize()
                errs[v] = float((out.float() - ref).norm() / ref.norm())
        C.lib().tsasr_gemm_set_nn128(0, 0, 0); C.lib().tsasr_gemm_set_plan(-1, 0)
        fl = 2.0 * M * N * K
        ingest = (128 + 64) * K * 2.0      # bytes one workgroup pulls through L2 -> LDS
        print(f"M={M} N={N} K={K}: {fl/1e9:.2f} GFLOP, {ingest/1e3:.0f} KB per workgroup through the LDS-DMA path; lib (yardstick) {libt:6.1f} us")
        for v in variants:
            tag = {0: "gemm ", 1: "FLOOR dma only", 2: "FLOOR mfma only"}[v[2]] + {0: "", 1: " wave-K", 2: " reg-pipe", 3: " loader waves"}[v[1]]
            extra = f"rel.err {errs[v]:.1e}" if v in errs else f"{ingest/best[v]/1e3:6.1f} GB/s per CU" if v[2] == 1 else ""
            print(f"   stages={v[0]} wavek={v[1]} {tag:26s} {best[v]:6.1f} us {fl/best[v]/1e6:7.1f} TF  {extra}")
    sys.exit(0)

for (M, N, K, ta, tb, f32) in SHAPES:
    A, B, out = operands(M, N, K, ta, tb, f32)
    mine = lambda: ops.gemm_bf16(A, B, M, N, K, M if ta else K, N if tb else K, ta, tb, out=out, accumulate=bool(f32))
    lib = lambda: torch.matmul(A.t() if ta else A, B if tb else B.t())
    best = {"mine": 1e9, "lib": 1e9}
    for rnd in range(2):   # interleaved rounds in one process; report the minimum
        best["mine"] = min(best["mine"], timeit(mine))
        best["lib"] = min(best["lib"], timeit(lib))
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + N * K) + (4.0 if f32 else 2.0) * M * N
    roof = max(by / 8e6, fl / 2.5e9)   # us: HBM 8 TB/s vs dense bf16 MFMA 2.5 PFLOP/s
    print(f"roof {roof:5.1f} us | M={M:6d} N={N:5d} K={K:6d} tA={ta} tB={tb} f32={f32}: ours {best['mine']:6.1f} us {fl/best['mine']/1e6:6.1f} TF | lib {best['lib']:6.1f} us {fl/best['lib']/1e6:6.1f} TF")

# fused epilogues at the FFN shapes (hot operands, graph-replayed): mode 1 = bias + LeakyReLU + dropout, mode 2 = mask/act' + colsum
M, F1, D = 8000, 2048, 256
x = torch.randn(M, D, device=DEV).to(torch.bfloat16); w1 = torch.randn(F1, D, device=DEV).to(torch.bfloat16)
b1 = torch.randn(F1, device=DEV); h = torch.randn(M, F1, device=DEV).to(torch.bfloat16)
do = torch.randn(M, D, device=DEV).to(torch.bfloat16); w2 = torch.randn(D, F1, device=DEV).to(torch.bfloat16)
dbias = torch.zeros(F1, device=DEV)
for name, fn in [
    ("fwd plain   N=2048 K=256", lambda: ops.gemm_bf16(x, w1, M, F1, D, D, D, 0, 0)),
    ("fwd fused<1> p=0.1      ", lambda: ops.gemm_bf16_fused(x, w1, M, F1, D, D, D, 0, 0, 1, bias=b1, slope=0.01, p=0.1, seed=5)),
    ("fwd fused<1> p=0        ", lambda: ops.gemm_bf16_fused(x, w1, M, F1, D, D, D, 0, 0, 1, bias=b1, slope=0.01, p=0.0, seed=5)),
    ("dgrad plain N=2048 K=256", lambda: ops.gemm_bf16(do, w2, M, F1, D, D, F1, 0, 1)),
    ("dgrad fused<2> p=0.1    ", lambda: ops.gemm_bf16_fused(do, w2, M, F1, D, D, F1, 0, 1, 2, y=h, slope=0.01, p=0.1, seed=5, dbias=dbias)),
    ("dgrad fused<2> p=0      ", lambda: ops.gemm_bf16_fused(do, w2, M, F1, D, D, F1, 0, 1, 2, y=h, slope=0.01, p=0.0, seed=5, dbias=dbias)),
]:
    print(f"{name}: {timeit(fn):6.1f} us")
