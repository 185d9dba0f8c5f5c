#!/usr/bin/env python3
"""Micro-benchmark of csrc/gemm.hip at the model's shapes (interleaved rounds in one process, each timing = 20 launches captured into one
hipGraph). `lib` = torch.matmul (hipBLASLt / rocBLAS) on the same operands: a YARDSTICK only, never dispatched by the package.
  python tools/gemm_bench.py            the shapes of the step, default kernels vs the library
  python tools/gemm_bench.py --floors   the 128x64 tile's lab floors (round-3 four-wave loop; its LDS-DMA ring alone; its fragment reads +
                                        MFMA alone) beside the product kernel (loader waves) at the N = 256 shapes of the step
  python tools/gemm_bench.py --chain    the FFN as the step runs it: up-projection -> down-projection, operand from beyond L2"""
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


if "--floors" in sys.argv:
    # lab floors of the 128x64 tile (csrc/gemm.hip gemm_nn128x64_lab_kernel): the round-3 four-wave loop, its DMA ring alone, its
    # fragment reads + MFMA alone, against the product kernel (loader waves) and the library - profiles/r04_notes.md section 1
    for (M, N, K) in [(8000, 256, 2048), (8000, 256, 768), (8000, 256, 256), (4000, 256, 2048)]:
        A, B, out = operands(M, N, K, 0, 0, 0)
        C.lib().tsasr_gemm_set_plan(1, 1)         # the 128x64 tile for every shape of this sweep
        ref = (A.float() @ B.float().t())
        fn = lambda: ops.gemm_bf16(A, B, M, N, K, K, K, 0, 0, out=out)
        modes = [(-1, "product kernel: 4 compute + 4 loader waves"), (0, "round-3 loop: 4 waves load AND compute"), (1, "FLOOR: its LDS-DMA ring alone"),
                 (2, "FLOOR: its fragment reads + MFMA alone")]
        best = {m: 1e9 for m, _ in modes}
        for rnd in range(3):
            for m, _ in modes:
                C.lib().tsasr_gemm_set_lab_floor(m)
                best[m] = min(best[m], timeit(fn))
        errs = {}
        for m in (-1, 0):
            C.lib().tsasr_gemm_set_lab_floor(m); out.zero_(); fn(); torch.cuda.synchronize()
            errs[m] = float((out.float() - ref).norm() / ref.norm())
        C.lib().tsasr_gemm_set_lab_floor(-1); C.lib().tsasr_gemm_set_plan(-1, 0)
        libt = min(timeit(lambda: torch.matmul(A, B.t())) for _ in range(2))
        fl = 2.0 * M * N * K
        ingest = (128 + 64) * K * 2.0      # bytes one workgroup pulls through L2 -> LDS
        print(f"M={M} N={N} K={K}: {fl/1e9:.2f} GFLOP, {ingest/1e3:.0f} KB per workgroup through the LDS-DMA path; library (yardstick only) {libt:6.1f} us")
        for m, tag in modes:
            extra = f"rel.err {errs[m]:.1e}" if m in errs else (f"{ingest/best[m]/1e3:6.1f} GB/s per CU" if m == 1 else "")
            print(f"   {tag:46s} {best[m]:6.1f} us {fl/best[m]/1e6:7.1f} TF  {extra}")
    sys.exit(0)

if "--chain" in sys.argv:
    # the FFN as the step runs it: up-projection (writes the 32.8 MB hidden activation) -> down-projection (reads it), 20 pairs per graph:
    # the down-projection's A operand comes from beyond L2, as in the step, not from an L2 that the previous replay left hot
    M, F1, D = 8000, 2048, 256
    x = torch.randn(M, D, device=DEV).to(torch.bfloat16); w1 = torch.randn(F1, D, device=DEV).to(torch.bfloat16)
    b1 = torch.randn(F1, device=DEV); w2 = torch.randn(D, F1, device=DEV).to(torch.bfloat16)
    hs = [torch.empty(M, F1, device=DEV, dtype=torch.bfloat16) for _ in range(4)]     # rotating buffers: 131 MB > L2, < Infinity Cache
    outs = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    state = {"i": 0}

    def up():
        h = hs[state["i"] % 4]
        C.check(C.lib().tsasr_gemm_bf16_fused(C.ptr(x), C.ptr(w1), C.ptr(h), M, F1, D, D, D, F1, 0, 0, 1, C.ptr(b1), None, 0, 0.01, 0.1, 5,
                                              None, None, None, None, 0, C.stream_ptr()), "up")
        return h

    def pair():
        h = up(); state["i"] += 1
        ops.gemm_bf16(h, w2, M, D, F1, F1, F1, 0, 0, out=outs)

    def down_hot():
        ops.gemm_bf16(hs[0], w2, M, D, F1, F1, F1, 0, 0, out=outs)

    def up_only():
        up(); state["i"] += 1
    res = [(timeit(up_only), timeit(down_hot), timeit(pair)) for _ in range(2)]
    u, d, p2 = (min(r[i] for r in res) for i in range(3))
    print(f"up-projection alone {u:5.1f} us | down-projection on a hot operand {d:5.1f} us | pair {p2:5.1f} us -> down-projection behind its producer {p2 - u:5.1f} us")
    # the same pairs with a DIFFERENT pair of weight matrices per launch (24 layers' worth, 48 MB + transposes: not L2-resident), as the
    # step's layers have
    w1s = [torch.randn(F1, D, device=DEV).to(torch.bfloat16) for _ in range(24)]
    w2s = [torch.randn(D, F1, device=DEV).to(torch.bfloat16) for _ in range(24)]

    def up_cold():
        h = hs[state["i"] % 4]
        C.check(C.lib().tsasr_gemm_bf16_fused(C.ptr(x), C.ptr(w1s[state["i"] % 24]), C.ptr(h), M, F1, D, D, D, F1, 0, 0, 1, C.ptr(b1), None, 0, 0.01, 0.1, 5,
                                              None, None, None, None, 0, C.stream_ptr()), "up")
        return h

    def pair_cold():
        h = up_cold()
        ops.gemm_bf16(h, w2s[state["i"] % 24], M, D, F1, F1, F1, 0, 0, out=outs)
        state["i"] += 1

    def up_cold_only():
        up_cold(); state["i"] += 1
    res = [(timeit(up_cold_only, 24), timeit(pair_cold, 24)) for _ in range(2)]
    u, p2 = (min(r[i] for r in res) for i in range(2))
    print(f"per-layer weights: up-projection alone {u:5.1f} us | pair {p2:5.1f} us -> down-projection behind its producer, cold weights {p2 - u:5.1f} us")
    # ... and a FRESH hidden activation per pair, as the step keeps every layer's for its backward: 24 x 32.8 MB = 787 MB, three times the
    # Infinity Cache - every write now costs HBM write bandwidth (a buffer rotating inside 131 MB is absorbed by the cache)
    del hs[:]
    hs.extend(torch.empty(M, F1, device=DEV, dtype=torch.bfloat16) for _ in range(24))

    def up_fresh():
        h = hs[state["i"] % 24]
        C.check(C.lib().tsasr_gemm_bf16_fused(C.ptr(x), C.ptr(w1s[state["i"] % 24]), C.ptr(h), M, F1, D, D, D, F1, 0, 0, 1, C.ptr(b1), None, 0, 0.01, 0.1, 5,
                                              None, None, None, None, 0, C.stream_ptr()), "up")
        return h

    def pair_fresh():
        h = up_fresh()
        ops.gemm_bf16(h, w2s[state["i"] % 24], M, D, F1, F1, F1, 0, 0, out=outs)
        state["i"] += 1

    def up_fresh_only():
        up_fresh(); state["i"] += 1
    res = [(timeit(up_fresh_only, 24), timeit(pair_fresh, 24)) for _ in range(2)]
    u, p2 = (min(r[i] for r in res) for i in range(2))
    print(f"fresh 32.8 MB activation per pair (787 MB in all): up-projection alone {u:5.1f} us | pair {p2:5.1f} us -> down-projection {p2 - u:5.1f} us")
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
