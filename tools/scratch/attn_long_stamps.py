"""Long-sequence attention forward alone (B=1, T=4000, causal) through the C ABI with the key-split workspace; phase stamps of the profile build (scratch)."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
PH = ("DMA issue", "wait group 0 + query fragments", "MFMAs (AC, G, PV) + G store", "skewed read, softmax, dropout", "last PV", "group waits / barriers", "barrier before merge", "merge + partial store")
def load(path):
    L = ctypes.CDLL(path)
    for name, (res, args) in C._PROTOS.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return L
B, T, H, Dh, causal, p = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 4, 64, 1, 0.1
D = H * Dh
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.5).to(DEV, torch.bfloat16)
pk = (torch.randn(2 * T - 1, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
u, v = (torch.randn(D, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
out, lse = torch.empty(B, T, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, T, device=DEV)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for path, prof in ((C.LIB_PATH, False), (os.path.join(root, "ts-asr_amd", "lib", "libtsasr_attnprof.so"), True)):
    if not os.path.exists(path):
        continue
    L = load(path)
    n = L.tsasr_relpos_attn_fwd_workspace_bytes(B, T, H)
    ws = torch.zeros(n, dtype=torch.uint8, device=DEV)
    def fwd():
        rc = L.tsasr_relpos_attn_fwd_ws(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, 1.0 / D ** 0.5, causal, p, 7, None,
                                        C.BF16, C.ptr(ws), n, C.stream_ptr())
        assert rc == 0
    for _ in range(3):
        fwd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fwd()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(path), f"fwd + merge {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    if prof:
        np_ = (T + 255) // 256
        ml = ws[B * H * T * np_ * 64 * 4:].view(torch.int64)[:8].cpu().tolist()
        tot = sum(ml)
        for name, c in zip(PH, ml):
            print(f"   {name:40s} {c:9d} cycles {100.0 * c / max(tot, 1):5.1f} %")
        print(f"   total {tot} cycles = {tot / 2400.0:.2f} us at 2.4 GHz (s_memtime ticks at 100 MHz: clock64 = shader clock)")
