import os, sys, time, importlib, torch
sys.path.insert(0, ".")
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
B, H, U = 1, 512, 1920
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
ref = None
for tag, env in (("groups", {"TSASR_LSTM_SEQ1": "0"}), ("seq1", {"TSASR_LSTM_SEQ1": "1"})):
    os.environ.pop("TSASR_LSTM_L2", None)
    os.environ.update(env)
    for rep in range(3):
        gates.copy_(gates0); torch.cuda.synchronize(); t0 = time.perf_counter()
        C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), nb, C.stream_ptr()), "fwd")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if "bwd" in sys.argv:
            C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nb, C.stream_ptr()), "bwd")
            torch.cuda.synchronize()
        t2 = time.perf_counter()
    err = int(ws[:8].view(torch.int32)[1])
    if ref is None: ref = (h.clone(), dgates.clone())
    print(f"{tag:8s} fwd {1e6*(t1-t0):7.0f} us  bwd {1e6*(t2-t1):7.0f} us  err {err}  h equal {torch.equal(h, ref[0])}  dgates rel {float((dgates.float()-ref[1].float()).norm()/ref[1].float().norm()) if 'bwd' in sys.argv else -1:.4f}", flush=True)
