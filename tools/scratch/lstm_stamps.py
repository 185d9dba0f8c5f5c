"""Per-phase cycle counts of the persistent LSTM kernels (lab build with -DLQ_PROFILE; scratch).
usage: python tools/scratch/lstm_stamps.py <lib.so> B U"""
import ctypes, sys, time
import torch
lib = ctypes.CDLL(sys.argv[1])
B, U, H = int(sys.argv[2]), int(sys.argv[3]), 512
lib.tsasr_lstm_seq_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
p = lambda t: ctypes.c_void_p(t.data_ptr())
g = torch.Generator(device="cpu").manual_seed(0)
gates0 = (torch.randn(B, U, H, 4, generator=g) * 0.5).to(dev)
whh = (torch.randn(4 * H, H, generator=g) * 0.04).to(dev).bfloat16()
whhT = whh.t().contiguous()
n = lib.tsasr_lstm_seq_workspace_bytes(B, U, H)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
c = torch.empty(B, U, H, device=dev); h = torch.empty(B, U, H, device=dev, dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    gates = gates0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.tsasr_lstm_seq_fwd(p(gates), p(c), p(h), p(whh), B, U, H, 1, p(ws), ctypes.c_size_t(n), st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stamps = ws[128:128 + 40].view(torch.int64).cpu().tolist()
    print(f"fwd rc={rc} {dt*1e6:.0f} us  per step: " + " ".join(f"{s / U:.0f}" for s in stamps), " (wait | load+mfma | cell | publish+drain | outputs)", flush=True)
dout = torch.randn(B, U, H, generator=g).to(dev).bfloat16()
dg = torch.empty(B, U, 4 * H, device=dev, dtype=torch.bfloat16)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.tsasr_lstm_seq_bwd(p(gates), p(c), p(dout), p(dg), p(whhT), B, U, H, 1, p(ws), ctypes.c_size_t(n), st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stamps = ws[128:128 + 40].view(torch.int64).cpu().tolist()
    print(f"bwd rc={rc} {dt*1e6:.0f} us  per step: " + " ".join(f"{s / U:.0f}" for s in stamps), " (wait | load+mfma | acc->lds | cell | publish+drain)", flush=True)
print("finite:", bool(torch.isfinite(h.float()).all()), bool(torch.isfinite(dg.float()).all()))
