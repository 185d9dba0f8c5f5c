import ctypes, importlib, os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
def run(B, T, pdrop, causal, ragged, H=4, Dh=64):
    L = C.lib()
    D = H * Dh
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    pk = (torch.randn(2 * T - 1, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    u, v = (torch.randn(D, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    if ragged: lens[-1] = max(1, T - 37); lens[0] = max(1, T // 2 + 1)
    dout = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    out, lse = torch.empty(B, T, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, T, device=DEV)
    dqkv, dpk = torch.zeros_like(qkv), torch.zeros_like(pk)
    du, dv = torch.zeros_like(u), torch.zeros_like(v)
    ws = torch.zeros(L.tsasr_relpos_attn_bwd_workspace_bytes(B, T, H), dtype=torch.uint8, device=DEV)
    kb = torch.zeros(max(L.tsasr_relpos_attn_keepbits_bytes(B, T, H), 16), dtype=torch.uint8, device=DEV)
    st, scale = C.stream_ptr(), 1.0 / D ** 0.5
    if pdrop > 0: L.tsasr_relpos_attn_keepbits(C.ptr(kb))
    assert L.tsasr_relpos_attn_fwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, scale, causal, pdrop, 7, None, C.BF16, st) == 0
    if pdrop > 0: L.tsasr_relpos_attn_keepbits(C.ptr(kb))
    assert L.tsasr_relpos_attn_bwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(dpk),
                                   C.ptr(du), C.ptr(dv), B, T, H, Dh, scale, causal, pdrop, 7, None, C.BF16, C.ptr(ws), ws.numel(), st) == 0
    torch.cuda.synchronize()
    return {"dqkv": dqkv.float().cpu(), "dpk": dpk.float().cpu(), "du": du.cpu(), "dv": dv.cpu()}
if len(sys.argv) > 1 and sys.argv[1] == "child":
    cases = [(2, 250, 0.0, 0, False), (3, 250, 0.1, 0, True), (2, 125, 0.1, 0, True), (2, 96, 0.1, 1, False), (2, 33, 0.0, 0, True), (2, 256, 0.2, 0, False), (2, 200, 0.1, 1, True)]
    torch.save([run(*c) for c in cases], sys.argv[2])
    sys.exit(0)
res = {}
for ver in ("2", "3"):
    path = f"/tmp/attn_cmp_{ver}.pt"
    subprocess.run([sys.executable, __file__, "child", path], env=dict(os.environ, TSASR_ATTN_SHORT=ver), check=True)
    res[ver] = torch.load(path)
names = ["B2 T250 p0", "B3 T250 p.1 ragged", "B2 T125 p.1 ragged", "B2 T96 p.1 causal", "B2 T33 p0 ragged", "B2 T256 p.2", "B2 T200 p.1 causal ragged"]
for n, a, b in zip(names, res["2"], res["3"]):
    def rel(x, y): return float((x - y).norm() / (y.norm() + 1e-20))
    D = 256
    dq = rel(b["dqkv"].view(*b["dqkv"].shape[:2], 4, 192)[..., :64], a["dqkv"].view(*a["dqkv"].shape[:2], 4, 192)[..., :64])
    dk = rel(b["dqkv"].view(*b["dqkv"].shape[:2], 4, 192)[..., 64:128], a["dqkv"].view(*a["dqkv"].shape[:2], 4, 192)[..., 64:128])
    dvv = rel(b["dqkv"].view(*b["dqkv"].shape[:2], 4, 192)[..., 128:], a["dqkv"].view(*a["dqkv"].shape[:2], 4, 192)[..., 128:])
    print(f"{n:28s} fused vs pair: dq {dq:.4f} dk {dk:.4f} dv {dvv:.4f} dpk {rel(b['dpk'], a['dpk']):.4f} du {rel(b['du'], a['du']):.4f} dv_bias {rel(b['dv'], a['dv']):.4f}   nan {bool(torch.isnan(b['dqkv']).any())}")
