#!/usr/bin/env python3
"""Phase split of relpos_attn_bwd_q (debug build -DAT_PROFILE): s_memtime deltas of wave 1 of workgroup (0,0,0)."""
import importlib, sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
C = importlib.import_module("ts-asr_amd._capi")
PROF = os.path.join(root, "ts-asr_amd", "lib", "libtsasr_hip_prof.so")   # prebuilt with -DAT_PROFILE (see the Makefile-free recipe below)
if not os.path.exists(PROF):
    subprocess.check_call(f"cd {root}/ts-asr_amd/csrc && mkdir -p /tmp/atprof && for f in *.hip capi.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -DAT_PROFILE -x hip -c $f -o /tmp/atprof/$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o {PROF} /tmp/atprof/*.o", shell=True)
C.LIB_PATH = PROF; C._lib = None
lib = C.lib()
B, T, H, Dh = 32, 250, 4, 64
D, R = H * Dh, 2 * T - 1
dev = "cuda"
g = torch.Generator().manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev).to(torch.bfloat16)
qkv, pk, dout = bf(B, T, 3 * D), bf(R, D), bf(B, T, D)
u, v = (torch.randn(H * Dh, generator=g) * 0.1).to(dev), (torch.randn(H * Dh, generator=g) * 0.1).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
out, lse = torch.empty(B, T, D, dtype=torch.bfloat16, device=dev), torch.empty(B, H, T, device=dev)
for _ in range(2):
    C.check(lib.tsasr_relpos_attn_fwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, 1 / 16.0, 0, 0.0, 0,
                                      None, C.BF16, C.stream_ptr()), "fwd")
torch.cuda.synchronize()
fst = lse.view(-1)[:14].view(torch.int64).cpu().tolist()
fnames = ["DMA issue + q loads", "wait staged tiles", "AC + G mfma, G store", "skew read / softmax / P", "P.V", "barrier before merge", "merge + epilogue"]
ftot = sum(fst)
print("fwd wave 1 of workgroup 0, cycles:")
for n, c in zip(fnames, fst):
    print(f"  {n:28s} {c:8d}  {100 * c / max(ftot, 1):5.1f}%")
print(f"  total {ftot} = {ftot / 2.4e3:.1f} us at 2.4 GHz")
C.check(lib.tsasr_relpos_attn_fwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, 1 / 16.0, 0, 0.0, 0,
                                  None, C.BF16, C.stream_ptr()), "fwd")
dqkv = torch.empty_like(qkv); dpk = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
du, dv = torch.empty_like(u), torch.empty_like(v)
ws = torch.zeros(lib.tsasr_relpos_attn_bwd_workspace_bytes(B, T, H), dtype=torch.uint8, device=dev)
for _ in range(2):
    C.check(lib.tsasr_relpos_attn_bwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(dpk),
                                      C.ptr(du), C.ptr(dv), B, T, H, Dh, 1 / 16.0, 0, 0.0, 0, None, C.BF16, C.ptr(ws), ws.numel(), C.stream_ptr()), "bwd")
torch.cuda.synchronize()
st = ws[:56].view(torch.int64).cpu().tolist()
names = ["prologue", "staging", "S/dP/G mfma + G store", "softmax/dS/skews + P,dS stores", "dQ mfma", "(sub-block end)", "epilogue"]
tot = sum(st)
print("bwd_q wave 1 of workgroup 0 (4 key tiles, 8 sub-blocks), cycles:")
for n, c in zip(names, st):
    print(f"  {n:28s} {c:8d}  {100 * c / tot:5.1f}%")
print(f"  total {tot} = {tot / 2.4e3:.1f} us at 2.4 GHz")
