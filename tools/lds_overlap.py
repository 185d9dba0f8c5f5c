#!/usr/bin/env python3
"""Narrowing down what the per-step LSTM kernels (csrc/lstm.hip lstm_step_fwd_kernel, 33280 B of static LDS) do to kernels that run beside
them (profiles/r03_notes.md): (1) eager, two free-running streams: LSTM steps on one, the Fbank on the other, device-side comparison with
the serial result; (2) LDS canaries of several sizes beside the LSTM steps (does any workgroup's LDS get written by somebody else?).
usage: python tools/lds_overlap.py [iters]"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

nnet = importlib.import_module("ts-asr_amd.nnet")
C = importlib.import_module("ts-asr_amd._capi")
from oracle.golden_recipe import golden_inputs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = "cuda"
inp = golden_inputs()
mix = torch.from_numpy(inp["mixed_sig"]).to(dev)
fb = nnet.Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).to(dev)
B, U, H = 4, 21, 128
gates0 = torch.randn(B, U, H, 4, device=dev)
whh = torch.randn(4 * H, H, device=dev).to(torch.bfloat16)
c = torch.empty(B, U, H, device=dev)
hh = torch.empty(B, U, H, dtype=torch.bfloat16, device=dev)
ws = torch.empty(C.lib().tsasr_lstm_seq_workspace_bytes(B, U, H), dtype=torch.uint8, device=dev)
gates = gates0.clone()


def lstm_steps(stream):
    C.check(C.lib().tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(hh), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), ws.numel(),
                                       ctypes.c_void_p(stream.cuda_stream)), "lstm")


ref = fb(mix).clone()
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(1, dtype=torch.int32, device=dev)
with torch.no_grad():
    for it in range(iters):
        lstm_steps(sa)
        with torch.cuda.stream(sb):
            out = fb(mix)
            bad.add_(((out != ref).sum() > 0).to(torch.int32))
torch.cuda.synchronize()
print(f"eager, two free-running streams (LSTM steps | Fbank): {int(bad.item())} of {iters} Fbank outputs wrong", flush=True)
bad.zero_()
with torch.no_grad():
    for it in range(iters):
        with torch.cuda.stream(sb):
            out = fb(mix)
            bad.add_(((out != ref).sum() > 0).to(torch.int32))
torch.cuda.synchronize()
print(f"eager, Fbank alone: {int(bad.item())} of {iters} wrong", flush=True)
err = torch.zeros(1, dtype=torch.int32, device=dev)
first = torch.zeros(4, dtype=torch.int32, device=dev)
for lds in (22544, 33280, 16384, 8192, 65536):
    err.zero_()
    first.zero_()
    for it in range(iters // 10):
        C.check(C.lib().tsasr_debug_lds_canary(1024, lds, 60, C.ptr(err), C.ptr(first), ctypes.c_void_p(sb.cuda_stream)), "canary")
        for _ in range(4):
            lstm_steps(sa)
    torch.cuda.synchronize()
    print(f"LDS canary ({lds} B per workgroup) beside LSTM steps: {int(err.item())} corrupted words; first {[hex(v & 0xffffffff) for v in first.tolist()]}", flush=True)

err2 = torch.zeros(2, dtype=torch.int32, device=dev)
for rounds in (8, 64):
    err2.zero_()
    first.zero_()
    for it in range(iters // 4):
        C.check(C.lib().tsasr_debug_barrier_canary(200, rounds, C.ptr(err2), C.ptr(first), ctypes.c_void_p(sb.cuda_stream)), "canary")
        lstm_steps(sa)
    torch.cuda.synchronize()
    print(f"barrier canary ({rounds} rounds, 200 workgroups) beside LSTM steps: {int(err2[0].item())} stale words; first {[hex(v & 0xffffffff) for v in first.tolist()]}", flush=True)
    err2.zero_()
    for it in range(iters // 4):
        C.check(C.lib().tsasr_debug_barrier_canary(200, rounds, C.ptr(err2), C.ptr(first), ctypes.c_void_p(sb.cuda_stream)), "canary")
    torch.cuda.synchronize()
    print(f"barrier canary ({rounds} rounds) alone: {int(err2[0].item())} stale words", flush=True)
