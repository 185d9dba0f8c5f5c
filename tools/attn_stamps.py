#!/usr/bin/env python3
"""Attention kernels alone, through the C ABI of one or more builds of the library (A/B of kernel changes inside one GPU lease, and
the in-kernel phase stamps of the -DAT_PROFILE build: `make -C ts-asr_amd/csrc prof`).

  python tools/attn_stamps.py [B T [pdrop]] [lib.so ...]      default: 32 250 0.1, the package's library + lib/libtsasr_attnprof.so

Per library: microseconds per forward call and per backward call (HIP events over `iters` back-to-back launches on one stream; the
backward call = query-major pass + key-major pass + d(pk) pass + its reduction), and for a profile build the share of each phase in
workgroup (0,0,0)'s cycles. Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations."""
import ctypes
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
FWD_PHASES = ("DMA issue (groups 0, 1)", "wait for group 0, query fragments", "DMA issue ahead, AC + G MFMAs, G store, P.V of the step before", "skewed read, softmax, dropout, P",
              "last P.V", "group waits + barriers of the steps", "barrier before the merge", "merge + epilogue")
BWD_PHASES = ("row loads, DMA issue, operand fragments", "wait for the tiles", "S, dP, G MFMAs + G store", "skewed read, p, dS", "P_d / dS stores, inverse skew", "dQ MFMAs",
              "merge of the key parts", "dQ store, bias partial sums")


def load(path):
    L = ctypes.CDLL(path)
    for name, (res, args) in C._PROTOS.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return L


def run(L, label, B, T, pdrop, H=4, Dh=64, iters=50, prof=False):
    D = H * Dh
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    pk = (torch.randn(2 * T - 1, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    u, v = (torch.randn(D, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    dout = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    out, lse = torch.empty(B, T, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, T, device=DEV)
    dqkv, dpk = torch.empty_like(qkv), torch.empty_like(pk)
    du, dv = torch.empty_like(u), torch.empty_like(v)
    ws = torch.empty(L.tsasr_relpos_attn_bwd_workspace_bytes(B, T, H), dtype=torch.uint8, device=DEV)
    kb_bytes = L.tsasr_relpos_attn_keepbits_bytes(B, T, H) if hasattr(L, "tsasr_relpos_attn_keepbits_bytes") and pdrop > 0 else 0
    kb = torch.empty(max(kb_bytes, 16), dtype=torch.uint8, device=DEV)
    st, scale = C.stream_ptr(), 1.0 / D ** 0.5

    def fwd():
        if kb_bytes:
            L.tsasr_relpos_attn_keepbits(C.ptr(kb))
        rc = L.tsasr_relpos_attn_fwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, scale, 0, pdrop, 7,
                                     None, C.BF16, st)
        assert rc == 0, L.tsasr_last_error()

    def bwd():
        if kb_bytes:
            L.tsasr_relpos_attn_keepbits(C.ptr(kb))
        rc = L.tsasr_relpos_attn_bwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(dpk),
                                     C.ptr(du), C.ptr(dv), B, T, H, Dh, scale, 0, pdrop, 7, None, C.BF16, C.ptr(ws), ws.numel(), st)
        assert rc == 0, L.tsasr_last_error()

    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        if name == "bwd":
            fwd()                      # a clean lse / keep-bits (the profile build's forward stamps clobber lse: rerun it unstamped? no - see below)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / iters * 1e3
    print(f"{label:28s} B={B} T={T} p={pdrop}: fwd {res['fwd']:7.1f} us   bwd {res['bwd']:7.1f} us   checksum dqkv {float(dqkv.float().abs().sum()):.6e} "
          f"dpk {float(dpk.float().abs().sum()):.6e}", flush=True)
    if prof:
        fwd()
        torch.cuda.synchronize()
        f = lse.view(-1)[:16].view(torch.int64).cpu().tolist()
        bwd()
        torch.cuda.synchronize()
        b_ = ws[:64].view(torch.int64).cpu().tolist()      # wave 0's row of the partial-sum slab
        for title, names, vals in (("forward (short kernel)", FWD_PHASES, f), ("backward query-major pass", BWD_PHASES, b_)):
            tot = max(1, sum(vals))
            print(f"  {title}: {tot} cycles in workgroup (0,0,0), wave 1 (forward: wave 0)")
            for n, c in zip(names, vals):
                print(f"    {100.0 * c / tot:5.1f} %  {c:8d}  {n}")
    return dqkv, dpk


if __name__ == "__main__":
    args = sys.argv[1:]
    nums = []
    while args and not args[0].endswith(".so"):
        nums.append(args.pop(0))
    B, T = (int(nums[0]), int(nums[1])) if len(nums) >= 2 else (32, 250)
    pdrop = float(nums[2]) if len(nums) >= 3 else 0.1
    libs = args or [C.LIB_PATH, os.path.join(os.path.dirname(C.LIB_PATH), "libtsasr_attnprof.so")]
    ref = None
    for path in libs:
        if not os.path.exists(path):
            print(f"{path}: not built, skipped")
            continue
        prof = "prof" in os.path.basename(path)
        got = run(load(path), os.path.basename(path), B, T, pdrop, prof=prof)
        if ref is not None and not prof:
            print("   same bits as the first library:", all(torch.equal(a, b) for a, b in zip(ref, got)))
        if ref is None:
            ref = got
