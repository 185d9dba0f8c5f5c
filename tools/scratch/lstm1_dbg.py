import os, sys, ctypes, importlib, torch
sys.path.insert(0, ".")
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
B, H, U = 1, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(U)
gates0 = (torch.randn(B, U, H, 4, generator=g) * 0.5).to(DEV)
whh = (torch.randn(4 * H, H, generator=g) * 0.04).to(DEV).to(torch.bfloat16)
whhT = whh.t().contiguous()
dout = torch.randn(B, U, H, generator=g).to(DEV).to(torch.bfloat16)
lib = C.lib()
nb = lib.tsasr_lstm_seq_workspace_bytes(B, U, H)
ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
gates, c = torch.empty_like(gates0), torch.empty(B, U, H, device=DEV)
h, dgates = torch.empty(B, U, H, dtype=torch.bfloat16, device=DEV), torch.empty(B, U, 4 * H, dtype=torch.bfloat16, device=DEV)
def run():
    gates.copy_(gates0)
    C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), nb, C.stream_ptr()), "fwd")
    C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nb, C.stream_ptr()), "bwd")
    torch.cuda.synchronize()
    return h.clone(), c.clone(), gates.clone(), dgates.clone()
os.environ["TSASR_LSTM_SEQ1"] = "0"; h0, c0, g0, d0 = run()
os.environ["TSASR_LSTM_SEQ1"] = "1"; h1, c1, g1, d1 = run()
print("h", torch.equal(h0, h1), "c", torch.equal(c0, c1), "gates", torch.equal(g0, g1))
for t in range(U - 1, max(U - 4, -1), -1):
    a, b = d0[0, t].float().view(4, H), d1[0, t].float().view(4, H)
    print("t", t, "rel", float((a - b).norm() / a.norm()), "per gate", [round(float((a[k] - b[k]).norm() / a[k].norm()), 4) for k in range(4)])
    if t == U - 2:
        e = ((a - b).abs() > 1e-3 * a.abs().max()).nonzero()
        print("  bad idx (gate, unit) first 20:", e[:20].tolist(), "count", len(e))
        print("  d0", a[0, :10].tolist()); print("  d1", b[0, :10].tolist())
