import importlib, sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
C = importlib.import_module("ts-asr_amd._capi")
if len(sys.argv) > 1:
    flag = sys.argv[1]
    subprocess.check_call(f"cd {root}/ts-asr_amd/csrc && mkdir -p /tmp/exp && for f in *.hip capi.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast {flag} -x hip -c $f -o /tmp/exp/$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/exp/libtsasr_hip.so /tmp/exp/*.o", shell=True)
    C.LIB_PATH = "/tmp/exp/libtsasr_hip.so"; C._lib = None
import torch
ops = importlib.import_module("ts-asr_amd.ops")
from tools.gemm_bench import timeit
for (M, N, K) in [(2048, 256, 8000)]:
    for ta, tb in ((1, 1), (0, 0), (0, 1)):
        A = torch.randn((K, M) if ta else (M, K), device="cuda").to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device="cuda").to(torch.bfloat16)
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32)
        fn = lambda: ops.gemm_bf16(A, B, M, N, K, M if ta else K, N if tb else K, ta, tb, out=out, accumulate=True)
        print(f"EXP [{M}x{N}] K={K} tA={ta} tB={tb}: {timeit(fn):6.1f} us", flush=True)
