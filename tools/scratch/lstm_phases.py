#!/usr/bin/env python3
"""Phase split of the persistent LSTM forward (debug build -DLQ_PROFILE: accumulated s_memtime deltas of one workgroup)."""
import importlib, sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
C = importlib.import_module("ts-asr_amd._capi")
subprocess.check_call(f"cd {root}/ts-asr_amd/csrc && mkdir -p /tmp/lqprof && for f in *.hip capi.cpp; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -DLQ_PROFILE -x hip -c $f -o /tmp/lqprof/$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lqprof/libtsasr_hip.so /tmp/lqprof/*.o", shell=True)
C.LIB_PATH = "/tmp/lqprof/libtsasr_hip.so"; C._lib = None
lib = C.lib()
B, U, H = 32, 121, 512
dev = "cuda"
gates = torch.randn(B, U, 4 * H, device=dev); c = torch.empty(B, U, H, device=dev); h = torch.empty(B, U, H, device=dev, dtype=torch.bfloat16)
whh = (torch.randn(4 * H, H, device=dev) * 0.05).to(torch.bfloat16)
ws = torch.zeros(lib.tsasr_lstm_seq_workspace_bytes(B, U, H), dtype=torch.uint8, device=dev)
for _ in range(3):
    C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), ws.numel(), C.stream_ptr()), "fwd")
torch.cuda.synchronize()
st = ws[128:128 + 40].view(torch.int64).cpu().tolist()
names = ["wait h", "loads+mfma", "cell", "publish+drain", "atomic"]
tot = sum(st)
print({n: f"{v / U:.0f} cyc/step ({100 * v / tot:.0f}%)" for n, v in zip(names, st)}, "total cyc/step", tot // U)
