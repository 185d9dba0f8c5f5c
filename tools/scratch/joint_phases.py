#!/usr/bin/env python3
"""Phase split of joint_bwd_y (debug build -DJY_PROFILE)."""
import importlib, sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
C = importlib.import_module("ts-asr_amd._capi")
subprocess.check_call(f"cd {root}/ts-asr_amd/csrc && mkdir -p /tmp/jyprof && for f in *.hip capi.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -DJY_PROFILE -x hip -c $f -o /tmp/jyprof/$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/jyprof/libtsasr_hip.so /tmp/jyprof/*.o", shell=True)
C.LIB_PATH = "/tmp/jyprof/libtsasr_hip.so"; C._lib = None
lib = C.lib()
B, T, U1, J, V, ldl = 32, 250, 121, 640, 29, 32
dev = "cuda"
enc = torch.randn(B, T, J, device=dev).to(torch.bfloat16); dec = torch.randn(B, U1, J, device=dev).to(torch.bfloat16)
W = torch.randn(V, J, device=dev) * 0.05
dl = torch.randn(B, T, U1, ldl, device=dev) * 0.01
denc = torch.empty_like(enc); ddec = torch.empty_like(dec); dW = torch.empty(V, J, device=dev); db = torch.empty(V, device=dev)
ws = torch.zeros(lib.tsasr_joint_bwd_workspace_bytes(B, T, U1, J), dtype=torch.uint8, device=dev)
import time
for _ in range(1):
    C.check(lib.tsasr_joint_bwd(C.ptr(dl), C.ptr(enc), C.ptr(dec), C.ptr(W), C.ptr(denc), C.ptr(ddec), C.ptr(dW), C.ptr(db), None, None,
                                B, T, U1, J, V, ldl, C.BF16, 0.01, C.ptr(ws), ws.numel(), C.stream_ptr()), "bwd")
torch.cuda.synchronize()
st = dec.view(torch.uint8).view(-1)[:24].view(torch.int64).cpu().tolist()
print("cycles of one workgroup's wave 0:", dict(zip(["dec tile loads", "dlogits loads", "enc + mfma + sums"], st)), "total", sum(st), f"= {sum(st)/2.38e3:.0f} us")
