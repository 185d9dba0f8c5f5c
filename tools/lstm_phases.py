#!/usr/bin/env python3
"""In-kernel phase times of the persistent LSTM kernels (csrc/lstm.hip built with -DLQ_PROFILE: s_memtime stamps of one workgroup,
left in the sync block of the workspace). usage (GPU box): tools/lstm_phases.sh ; prints cycles per step and phase for B, U."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
C = importlib.import_module(entry.PKG + "._capi")
B, U, H = int(sys.argv[1]), int(sys.argv[2]), 512
dev = "cuda:0"
lib, st = C.lib(), C.stream_ptr()
g = torch.Generator().manual_seed(0)
gates = (torch.randn(B, U, H, 4, generator=g) * 0.5).to(dev)
c = torch.empty(B, U, H, device=dev)
h = torch.empty(B, U, H, dtype=torch.bfloat16, device=dev)
whh = (torch.randn(4 * H, H, generator=g) * 0.04).to(torch.bfloat16).to(dev)
nbytes = lib.tsasr_lstm_seq_workspace_bytes(B, U, H)
assert lib.tsasr_lstm_seq_persistent(B, H, C.BF16)
for name in ("fwd", "bwd"):
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    if name == "fwd":
        C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), nbytes, st), "fwd")
        labels = ["wait h(t-1)", "exchange loads + LDS + MFMA", "cell math", "publish + drain", "arrive + stores"]
    else:
        dout = (torch.randn(B, U, H, generator=g) * 0.1).to(torch.bfloat16).to(dev)
        dgates = torch.empty(B, U, 4 * H, dtype=torch.bfloat16, device=dev)
        whhT = whh.t().contiguous()
        C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nbytes, st), "bwd")
        labels = ["input loads issued + wait dgates(t+1)", "exchange loads + MFMA issue", "accumulators to LDS + barrier", "cell backward + barrier", "publish + drain"]
    ev1.record()
    torch.cuda.synchronize()
    t = ws[128:168].view(torch.int64).cpu().tolist()
    print(f"{name}: B={B} U={U}: {ev0.elapsed_time(ev1) * 1e3 / U:.2f} us per step; stamp ticks per step (100 MHz s_memtime on gfx950 = 10 ns each):")
    for lab, v in zip(labels, t):
        print(f"   {lab:42s} {v / U:9.1f}")
