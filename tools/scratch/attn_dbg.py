import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
C = importlib.import_module("ts-asr_amd._capi")
DEV = "cuda:0"
def ref(qkv, pk, u, v, lens, H, scale, causal, variant=0):
    B, T, D3 = qkv.shape; D = D3 // 3; Dh = D // H
    x = qkv.float().view(B, T, H, 3 * Dh)
    q, k, vv = x[..., :Dh], x[..., Dh:2 * Dh], x[..., 2 * Dh:]
    p = pk.float().view(2 * T - 1, H, Dh)
    uu, vb = u.view(H, Dh), v.view(H, Dh)
    bf = lambda t: t.to(torch.bfloat16).float()
    ac = torch.einsum("bihd,bjhd->bhij", bf(q + uu), k)
    g = torch.einsum("bihd,rhd->bhir", bf(q + vb), p)
    idx = torch.arange(T, device=qkv.device)
    rr = idx[None, :] - idx[:, None] + T - 1
    bd = torch.gather(g, 3, rr[None, None].expand(B, H, T, T))
    if variant == 1: bd = bd * 0
    if variant == 2: ac = torch.einsum("bihd,bjhd->bhij", bf(q), k); 
    if variant == 3: bd = torch.gather(torch.einsum("bihd,rhd->bhir", bf(q), p), 3, rr[None, None].expand(B, H, T, T))
    if variant == 4:
        bd = bd.clone(); t = bd[..., 0::2].clone(); bd[..., 0::2] = bd[..., 1::2]; bd[..., 1::2] = t
    if variant == 5: bd = torch.gather(g, 3, (rr[None, None] + 1).clamp(max=2 * T - 2).expand(B, H, T, T))
    if variant == 6: bd = torch.gather(g, 3, (rr[None, None] - 1).clamp(min=0).expand(B, H, T, T))
    jj = idx[None, :].expand(T, T); ii = idx[:, None].expand(T, T)
    if variant == 7: bd = bd * (jj % 2 == 0)
    if variant == 8: bd = bd * (jj % 2 == 1)
    if variant == 9: bd = bd * (((jj % 32) - (ii % 32)) <= 0)
    if variant == 10: bd = bd * (((jj % 32) - (ii % 32)) > 0)
    if variant == 11: bd = bd * ((jj % 8) < 4)
    if variant == 12: bd = bd * ((jj % 8) >= 4)
    if variant == 13: bd = torch.gather(torch.einsum("bihd,rhd->bhir", bf(q + uu), p), 3, rr[None, None].expand(B, H, T, T))
    sc = (ac + bd) * scale
    if causal: sc = sc.masked_fill(idx[None, :] > idx[:, None], float("-inf"))
    sc = sc.masked_fill((idx[None, :] >= lens[:, None]).view(B, 1, 1, T), float("-inf"))
    pr = torch.softmax(sc, -1)
    return torch.einsum("bhij,bjhd->bihd", pr, vv).reshape(B, T, D), torch.logsumexp(sc, -1)
def run(B, T, H=4, Dh=64, causal=0, ragged=False, zu=False, zv=False, zp=False):
    D = H * Dh
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    pk = (torch.randn(2 * T - 1, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    u, v = (torch.randn(D, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    if zu: u = u * 0
    if zv: v = v * 0
    if zp: pk = pk * 0
    print("zu zv zp", zu, zv, zp)
    lens = torch.full((B,), T, dtype=torch.int32, device=DEV)
    if ragged: lens[-1] = max(1, T - 37)
    out, lse = torch.zeros(B, T, D, dtype=torch.bfloat16, device=DEV), torch.zeros(B, H, T, device=DEV)
    L = C.lib()
    rc = L.tsasr_relpos_attn_fwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, T, H, Dh, 1.0 / D ** 0.5, causal, 0.0, 7, None, C.BF16, C.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    for var in (1, 7, 8, 9, 10, 11, 12, 13):
        ro, rl = ref(qkv, pk, u, v, lens, H, 1.0 / D ** 0.5, causal, var)
        print(f"      variant {var}: rel {float((out.float() - ro).norm() / ro.norm()):.4f}")
    ro, rl = ref(qkv, pk, u, v, lens, H, 1.0 / D ** 0.5, causal)
    r1, _ = ref(qkv, pk, u, v, lens, H, 1.0 / D ** 0.5, causal, 1)
    print("      |true - noBD| rel", float((ro - r1).norm() / ro.norm()))
    e = (out.float() - ro).view(B, T, H, Dh)
    print(f"B={B} T={T} causal={causal} ragged={ragged}: rel {float(e.norm() / ro.norm()):.4f}  lse maxerr {float((lse - rl).abs().max()):.4f}")
    per_row = e.pow(2).sum((0, 2, 3)).sqrt().cpu()
    bad = (per_row > 0.05 * per_row.max()).nonzero().flatten().tolist()
    print("   bad rows:", bad[:40], "..." if len(bad) > 40 else "", len(bad))
    per_h = e.pow(2).sum((0, 1, 3)).sqrt().cpu().tolist(); print("   per head:", [round(x, 3) for x in per_h])
    per_d = e.pow(2).sum((0, 1, 2)).sqrt().cpu(); print("   per dim :", [round(float(x), 2) for x in per_d[::4]])
    le = (lse - rl).abs().amax((0, 1)).cpu(); print("   lse err rows:", (le > 1e-2).nonzero().flatten().tolist()[:40])
for a in ((1, 32), (1, 250)):
    run(*a)
