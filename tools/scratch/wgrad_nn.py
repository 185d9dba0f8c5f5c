import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
ops = importlib.import_module("ts-asr_amd.ops"); C = importlib.import_module("ts-asr_amd._capi")
from tools.gemm_bench import timeit
DEV = "cuda"
for (M, N, K) in [(2048, 256, 8000), (256, 2048, 8000), (768, 256, 8000), (256, 256, 8000)]:
    for ta, tb in ((1, 1), (0, 0), (0, 1), (1, 0)):
        A = torch.randn((K, M) if ta else (M, K), device=DEV).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device=DEV).to(torch.bfloat16)
        out = torch.zeros(M, N, device=DEV, dtype=torch.float32)
        fn = lambda: ops.gemm_bf16(A, B, M, N, K, M if ta else K, N if tb else K, ta, tb, out=out, accumulate=True)
        print(f"[{M}x{N}] K={K} tA={ta} tB={tb}: {timeit(fn):6.1f} us", flush=True)
