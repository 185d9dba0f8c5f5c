"""GPU parity of the Conformer block kernels (through the C-ABI) vs golden vectors produced by the reference
(tests/golden/c1_blocks.npz: RelPosMHAXL, ConvolutionModule, FFN, ConformerEncoderLayer - outputs AND gradients) and vs
plain fp32 torch formulas for the row kernels."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.golden_recipe import CFG1, det_tensor, load_det_weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nn_():
    m = importlib.import_module("ts-asr_amd.nnet")
    m.set_compute_dtype(torch.float32)
    return m


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("ts-asr_amd.ops")


@pytest.fixture
def mfma_fp32(monkeypatch):
    """fp32 STORAGE through the MFMA kernels (operands rounded to bf16 inside): what fp32 tensors reached before the exact-fp32 attention /
    joint kernels became the fp32 path (ops.ATTN_F32_EXACT, rnnt.JOINT_F32_EXACT); these instantiations stay covered."""
    monkeypatch.setattr(importlib.import_module("ts-asr_amd.ops"), "ATTN_F32_EXACT", False)
    monkeypatch.setattr(importlib.import_module("ts-asr_amd.rnnt"), "JOINT_F32_EXACT", False)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def close(a, b, atol, rtol=1e-3):
    if isinstance(b, torch.Tensor):
        b = b.detach().float().cpu().numpy()
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), np.asarray(b), atol=atol, rtol=rtol)


def block_inputs(golden):
    g = golden["c1_blocks"]
    x = T(g["x"]).to(DEV).requires_grad_(True)
    lens = (T(np.asarray(CFG1["mix_lens"], np.float32)) * 50).round().to(torch.int32).to(DEV)
    probe = T(det_tensor("probe.blk", (4, 50, 144), 1.0)).to(DEV)
    return g, x, lens, probe


def check_grads(mod, g, tag, atol=3e-4, rtol=2e-3, skip=()):
    n = 0
    for name, p in mod.named_parameters():
        key = f"{tag}:d.{name}"
        if key in g.files and name not in skip:
            close(p.grad, g[key], atol, rtol)
            n += 1
    return n


# ---------------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("M,D", [(200, 144), (8000, 256), (37, 1024), (50, 2560), (33, 5120)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("slope", [None, 0.01])
def test_layernorm(ops, M, D, dtype, slope):
    g = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(dtype)
    w, b = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    dy = torch.randn(M, D, generator=g).to(dtype)
    xr = x.float().clone().requires_grad_()
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.layer_norm(xr, (D,), wr, br, 1e-5)
    if slope is not None:
        yr = F.leaky_relu(yr, slope)
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_()
    wg, bg = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.layer_norm(xg, wg, bg, 1e-5, slope)
    y.backward(dy.to(DEV))
    lo = dtype == torch.bfloat16
    close(y, yr.detach(), 3e-2 if lo else 2e-5, 1e-2 if lo else 1e-4)
    close(xg.grad, xr.grad, 6e-2 if lo else 1e-4, 2e-2 if lo else 1e-3)
    close(wg.grad, wr.grad, (0.5 if lo else 2e-3) * (M / 200) ** 0.5, 2e-2 if lo else 1e-3)
    close(bg.grad, br.grad, (0.5 if lo else 2e-3) * (M / 200) ** 0.5, 2e-2 if lo else 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_act_dropout(ops, dtype):
    M, N = 500, 2048
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, N, generator=g).to(dtype)
    b = torch.randn(N, generator=g) * 0.1
    # p = 0: exact formula
    xg, bg = x.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.bias_act_dropout(xg, bg, 0.01, 0.0, True)
    ref = F.leaky_relu(x.float() + b, 0.01)
    close(y, ref, 2e-2 if dtype == torch.bfloat16 else 1e-6)
    dy = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    y.backward(dy)
    dref = dy.float().cpu() * torch.where(ref > 0, 1.0, 0.01)
    close(xg.grad, dref, 2e-2 if dtype == torch.bfloat16 else 1e-6)
    close(bg.grad, dref.sum(0), 0.3 if dtype == torch.bfloat16 else 1e-3, 2e-2)
    # p = 0.1: kept fraction, scaling, and backward regenerates the very same mask
    xg = x.to(DEV).requires_grad_()
    y = ops.bias_act_dropout(xg, None, None, 0.1, True)
    keep = (y != 0)
    frac = keep.float().mean().item()
    assert abs(frac - 0.9) < 3e-3
    close(y[keep], (x.to(DEV).float() / 0.9)[keep], 2e-2 if dtype == torch.bfloat16 else 1e-6)
    y.backward(torch.ones_like(y))
    close(xg.grad, keep.float() / 0.9, 1e-2 if dtype == torch.bfloat16 else 1e-6)
    # eval mode: identity (+bias)
    close(ops.bias_act_dropout(x.to(DEV), None, None, 0.1, False), x.float(), 0)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_stream_statistics(ops, p):
    """The counter-based mask (csrc/common.h drop_hash): keep rate, no correlation between neighbours, between the two halves of
    one hash word, between rows, or between the streams of consecutive call seeds."""
    M, N = 2048, 1024
    ones = torch.ones(M, N, device=DEV, dtype=torch.float32)
    masks = [(ops._BiasActDropoutFn.apply(ones, None, -1.0, p, seed) != 0).float() for seed in (1000, 1001, 1002 + (1 << 32))]
    n = M * N
    sig = (p * (1 - p) / n) ** 0.5
    for m in masks:
        assert abs(m.mean().item() - (1 - p)) < 5 * sig + 1e-5     # 1e-5: p is quantised to 1/65536
        assert abs(m[:, 0::2].mean().item() - m[:, 1::2].mean().item()) < 8 * sig

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).mean() / (a.std() * b.std()))

    lim = 6 / n ** 0.5
    m = masks[0]
    assert abs(corr(m[:, :-1], m[:, 1:])) < lim          # neighbours (odd pairs share a hash word)
    assert abs(corr(m[:, 0::2], m[:, 1::2])) < lim * 1.5
    assert abs(corr(m[:-1], m[1:])) < lim                # rows
    assert abs(corr(masks[0], masks[1])) < lim           # consecutive seeds
    assert abs(corr(masks[1], masks[2])) < lim
    assert abs(corr(masks[0][:, :-2], masks[1][:, 2:])) < lim   # shifted streams


def test_dropout_add_and_time_mask(ops):
    B, Tn, N = 3, 17, 64
    g = torch.Generator().manual_seed(2)
    x, res = torch.randn(B, Tn, N, generator=g), torch.randn(B, Tn, N, generator=g)
    b = torch.randn(N, generator=g)
    lens = torch.tensor([17, 9, 1], dtype=torch.int32)
    xg, rg, bg = x.to(DEV).requires_grad_(), res.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    out = ops.dropout_add(xg, bg, rg, 0.5, 0.0, True, lens.to(DEV))
    mask = (torch.arange(Tn)[None, :] < lens[:, None]).float().unsqueeze(-1)
    ref = res + 0.5 * (x + b) * mask
    close(out, ref, 1e-6)
    dout = torch.randn(B, Tn, N, generator=g)
    out.backward(dout.to(DEV))
    close(xg.grad, 0.5 * dout * mask, 1e-6)
    close(rg.grad, dout, 0)
    close(bg.grad, (0.5 * dout * mask).sum((0, 1)), 1e-5)
    # broadcast residual [B,1,N] (speaker embedding injection)
    r1 = torch.randn(B, 1, N, generator=g).to(DEV).requires_grad_()
    o2 = ops.dropout_add(xg.detach(), None, r1)
    o2.sum().backward()
    close(o2, x + r1.detach().cpu(), 1e-6)
    close(r1.grad, torch.full((B, 1, N), float(Tn)), 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,p,lens", [(256, 0.0, True), (256, 0.1, True), (144, 0.1, False), (1024, 0.0, False)])
def test_add_layer_norm_equals_dropout_add_then_layernorm(ops, dtype, D, p, lens):
    """The fused seam (s, LN(s)) against the two tested single kernels run with the same dropout seed, forward and backward."""
    B, Tn = 3, 21
    g = torch.Generator().manual_seed(D)
    mk = lambda *sh: torch.randn(*sh, generator=g)
    x, res, b, gam, bet = mk(B, Tn, D), mk(B, Tn, D), mk(D), 1 + 0.1 * mk(D), 0.1 * mk(D)
    ds, dy = mk(B, Tn, D), mk(B, Tn, D)
    vl = torch.tensor([21, 9, 1], dtype=torch.int32).to(DEV) if lens else None
    trows = Tn if lens else 0
    seed = 1234 if p > 0 else 0

    def leaves():
        return [t.to(DEV).to(dtype if i < 2 else torch.float32).requires_grad_() for i, t in enumerate((x, res, b, gam, bet))]

    a = leaves()
    s1, y1 = ops._AddLayerNormFn.apply(a[0], a[2], a[1], a[3], a[4], 0.5, p, seed, vl, trows, 1e-5)
    torch.autograd.backward([s1, y1], [ds.to(DEV).to(dtype), dy.to(DEV).to(dtype)])
    r = leaves()
    s2 = ops._DropoutAddFn.apply(r[0], r[2], r[1], 0.5, p, seed, vl, trows)
    y2 = ops._LayerNormFn.apply(s2, r[3], r[4], 1e-5, -1.0)
    torch.autograd.backward([s2, y2], [ds.to(DEV).to(dtype), dy.to(DEV).to(dtype)])
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    close(s1, s2, 0 if dtype == torch.float32 else tol)
    close(y1, y2, tol)
    for u, v, name in zip(a, r, ("dx", "dres", "dbias", "dgamma", "dbeta")):
        scale = max(1.0, float(v.grad.abs().max()))
        close(u.grad, v.grad, tol * scale, rtol=1e-4 if dtype == torch.float32 else 2e-2)
    # y unused downstream (last seam of a layer returns only LN(s)): gradient flows through y alone
    a = leaves()
    _, y3 = ops._AddLayerNormFn.apply(a[0], a[2], a[1], a[3], a[4], 0.5, p, seed, vl, trows, 1e-5)
    y3.backward(dy.to(DEV).to(dtype))
    r = leaves()
    y4 = ops._LayerNormFn.apply(ops._DropoutAddFn.apply(r[0], r[2], r[1], 0.5, p, seed, vl, trows), r[3], r[4], 1e-5, -1.0)
    y4.backward(dy.to(DEV).to(dtype))
    close(a[1].grad, r[1].grad, tol, rtol=1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("M,Tn,p,lens,conv_w", [(8000, 250, 0.1, True, False), (8000, 250, 0.0, False, True), (4000, 125, 0.1, False, False),
                                               (77, 11, 0.1, True, True), (32, 32, 0.0, False, False), (1, 1, 0.1, True, False)])
def test_linear_add_layer_norm_equals_gemm_then_row_kernel(ops, M, Tn, p, lens, conv_w):
    """csrc/linear_ln.hip (projection + residual tail + LayerNorm in one launch, N = K = 256) against the GEMM and the row kernel it replaces,
    same dropout seed: (s, y) and every gradient bit for bit - full tiles, a ragged last tile, fewer rows than a tile, a Conv1d(k=1) weight."""
    D = 256
    B = M // Tn
    g = torch.Generator().manual_seed(M + 7)
    mk = lambda *sh: torch.randn(*sh, generator=g)
    xin, res, w, b, gam, bet = mk(B, Tn, D), mk(B, Tn, D), mk(D, D) / 16, mk(D), 1 + 0.1 * mk(D), 0.1 * mk(D)
    if conv_w:
        w = w.unsqueeze(-1)
    ds, dy = mk(B, Tn, D), mk(B, Tn, D)
    vl = torch.randint(1, Tn + 1, (B,), generator=g, dtype=torch.int32).to(DEV) if lens else None
    trows = Tn if lens else 0
    seed = 4321 if p > 0 else 0

    def leaves():
        return [t.to(DEV).to(torch.bfloat16 if i < 2 else torch.float32).requires_grad_() for i, t in enumerate((xin, res, w, b, gam, bet))]

    a = leaves()
    assert ops._gemm_ok(a[0], a[2])
    s1, y1 = ops._LinearAddLayerNormFn.apply(a[0], a[2], a[3], a[1], a[4], a[5], 0.5, p, seed, vl, trows, 1e-5)
    torch.autograd.backward([s1, y1], [ds.to(DEV).bfloat16(), dy.to(DEV).bfloat16()])
    r = leaves()
    s2, y2 = ops._AddLayerNormFn.apply(ops._LinearFn.apply(r[0], r[2]), r[3], r[1], r[4], r[5], 0.5, p, seed, vl, trows, 1e-5)
    torch.autograd.backward([s2, y2], [ds.to(DEV).bfloat16(), dy.to(DEV).bfloat16()])
    assert torch.equal(s1, s2) and torch.equal(y1, y2)
    for u, v, name in zip(a, r, ("dxin", "dres", "dW", "dbias", "dgamma", "dbeta")):
        assert torch.equal(u.grad, v.grad), name
    # and the public entry takes the fused path for this shape (the switch falls back to the pair)
    ln = torch.nn.LayerNorm(D).to(DEV)
    s3, y3 = ops.linear_add_layer_norm(a[0].detach(), a[2].detach(), a[3].detach(), a[1].detach(), ln, 1.0, 0.0, False, vl)
    old, ops.LINEAR_LN_FUSED = ops.LINEAR_LN_FUSED, False
    try:
        s4, y4 = ops.linear_add_layer_norm(a[0].detach(), a[2].detach(), a[3].detach(), a[1].detach(), ln, 1.0, 0.0, False, vl)
    finally:
        ops.LINEAR_LN_FUSED = old
    assert torch.equal(s3, s4) and torch.equal(y3, y4)


# ---------------------------------------------------------------------------------------------- blocks vs reference golden
@pytest.mark.parametrize("tag,causal", [("conv", False), ("conv_causal", True)])
def test_convolution_module_vs_reference(nn_, golden, tag, causal):
    g, x, lens, probe = block_inputs(golden)
    m = load_det_weights(nn_.ConvolutionModule(144, 31, True, torch.nn.LeakyReLU, 0.0, causal=causal), "blk.conv.").to(DEV)
    out = m(x, valid_lens=lens)
    (out * probe).sum().backward()
    close(out, g[f"{tag}:out"], 1e-4)
    close(x.grad, g[f"{tag}:dx"], 3e-4, 2e-3)
    assert check_grads(m, g, tag) == 10


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_convmod_core_full_width(ops, dtype, causal):
    """D=256, T=250 (BASELINE configs[1] layer shape), ragged tile edge (250 = 7*32 + 26): HIP vs torch fp32 formula."""
    B, Tn, D, K = 3, 250, 256, 31
    g = torch.Generator().manual_seed(5)
    y2 = torch.randn(B, Tn, 2 * D, generator=g).to(dtype)
    b2, cw, cb = torch.randn(2 * D, generator=g) * 0.1, torch.randn(D, 1, K, generator=g) / K ** 0.5, torch.randn(D, generator=g) * 0.1
    lw, lb = torch.randn(D, generator=g) * 0.1 + 1, torch.randn(D, generator=g) * 0.1
    dz = torch.randn(B, Tn, D, generator=g).to(dtype)
    ps = [t.clone().requires_grad_() for t in (y2.float(), b2, cw, cb, lw, lb)]
    h = ps[0] + ps[1]
    gl = (h[..., :D] * torch.sigmoid(h[..., D:])).transpose(1, 2)
    gl = F.pad(gl, (K - 1, 0)) if causal else F.pad(gl, (K // 2, K // 2))
    c = F.conv1d(gl, ps[2], ps[3], groups=D).transpose(1, 2)
    zr = F.leaky_relu(F.layer_norm(c, (D,), ps[4], ps[5], 1e-5), 0.01)
    zr.backward(dz.float())
    pg = [t.to(DEV).requires_grad_() for t in (y2, b2, cw, cb, lw, lb)]
    z = ops.convmod_core(*pg, causal, 1e-5, 0.01)
    z.backward(dz.to(DEV))
    lo = dtype == torch.bfloat16
    close(z, zr.detach(), 4e-2 if lo else 1e-4, 2e-2 if lo else 1e-3)
    for a, r, name in zip(pg, ps, ("dy2", "db2", "dcw", "dcb", "dlnw", "dlnb")):
        rel = float((a.grad.float().cpu() - r.grad).norm() / r.grad.norm())
        assert rel < (2e-2 if lo else 2e-4), (name, rel)


@pytest.mark.parametrize("B,Tn,K,causal,bias", [(3, 250, 31, False, True), (2, 125, 31, True, True), (1, 31, 31, False, False), (2, 77, 15, True, True),
                                                (1, 1000, 31, True, False), (4, 33, 7, False, True), (2, 64, 3, False, True)])
def test_convmod_one_launch_matches_pair(ops, monkeypatch, B, Tn, K, causal, bias):
    """bf16 rows of D = 256: the one-launch-per-direction kernels (csrc/convmod.hip, convmod_*_fused_kernel) against the two / three-launch pair
    they replace, same inputs: z, the saved statistics' consequences and dy2 BIT FOR BIT (c and dc are rounded where the pair stored them, the row
    sums use the same lanes in the same order); the parameter gradients (summed per workgroup in another order, fp32) to 2e-5 relative."""
    D = 256
    g = torch.Generator().manual_seed(B * 1000 + Tn + K)
    y2 = torch.randn(B, Tn, 2 * D, generator=g).bfloat16()
    b2 = torch.randn(2 * D, generator=g) * 0.1 if bias else None
    cw, cb = torch.randn(D, 1, K, generator=g) / K ** 0.5, torch.randn(D, generator=g) * 0.1
    lw, lb = torch.randn(D, generator=g) * 0.1 + 1, torch.randn(D, generator=g) * 0.1
    dz = torch.randn(B, Tn, D, generator=g).bfloat16().to(DEV)

    def run(flag):
        monkeypatch.setenv("TSASR_CONVMOD_FUSED", flag)
        pg = [None if t is None else t.to(DEV).requires_grad_() for t in (y2, b2, cw, cb, lw, lb)]
        z = ops.convmod_core(*pg, causal, 1e-5, 0.01)
        z.backward(dz)
        torch.cuda.synchronize()
        return z.detach(), [None if t is None else t.grad for t in pg]

    z1, g1 = run("1")
    z0, g0 = run("0")
    assert torch.equal(z1, z0)
    assert torch.equal(g1[0], g0[0]), float((g1[0].float() - g0[0].float()).abs().max())
    for a, r, name in zip(g1[1:], g0[1:], ("db2", "dcw", "dcb", "dlnw", "dlnb")):
        if a is None:
            continue
        rel = float((a - r).norm() / r.norm())
        assert rel < 2e-5, (name, rel)


def test_ffn_and_layer_vs_reference(nn_, golden):
    g, x, lens, probe = block_inputs(golden)
    for tag, causal in (("layer", False), ("layer_causal", True)):
        x.grad = None
        lay = load_det_weights(nn_.ConformerEncoderLayer(d_model=144, d_ffn=576, nhead=4, kernel_size=31, activation=torch.nn.LeakyReLU,
                                                         dropout=0.0, causal=causal), "blk.layer.").to(DEV)
        pe = nn_.RelPosEncXL(144).to(DEV)(x)
        out, _ = lay(x, pos_embs=pe, valid_lens=lens, need_attn=False)
        (out * probe).sum().backward()
        # the layer's attention runs on bf16 MFMA operands even when activations are fp32: relative L2 is the yardstick
        def rel(a, b):
            b = T(b)
            return float((a.detach().float().cpu() - b).norm() / b.norm())
        assert rel(out, g[f"{tag}:out"]) < 5e-3
        close(out, g[f"{tag}:out"], 3e-2, 2e-2)
        assert rel(x.grad, g[f"{tag}:dx"]) < 2e-2
        n = 0
        for name, p in lay.named_parameters():
            key = f"{tag}:d.{name}"
            if key in g.files:
                assert rel(p.grad, g[key]) < 3e-2, (name, rel(p.grad, g[key]))
                n += 1
        assert n > 20
    # macaron half step alone: x + 0.5*FFN(x); golden holds FFN(x) and its grads
    x2 = T(g["x"]).to(DEV).requires_grad_(True)
    y = lay._ffn_add(x2, lay.ffn_module1)
    close((y - x2) * 2, g["ffn:out"], 1e-4)


@pytest.mark.parametrize("tag,use_lens,causal", [("mha_nomask", False, False), ("mha_kpm", True, False), ("mha_kpm_causal", True, True)])
def test_relpos_mha_vs_reference(nn_, golden, tag, use_lens, causal):
    g, x, lens, probe = block_inputs(golden)
    m = load_det_weights(nn_.RelPosMHAXL(144, 4, dropout=0.0, mask_pos_future=causal), "blk.mha.").to(DEV)
    pe = nn_.RelPosEncXL(144).to(DEV)(x)
    close(pe, golden["c1_blocks"]["relpos_table"], 1e-6)
    out, attn = m(x, x, x, pe, key_lens=lens if use_lens else None, causal=causal, return_attn_weights=True)
    (out * probe).sum().backward()
    close(out, g[f"{tag}:out"], 1e-4)
    close(x.grad, g[f"{tag}:dx"], 3e-4, 2e-3)
    assert check_grads(m, g, tag) == 6
    if tag == "mha_kpm":
        close(attn[1], g["mha_kpm:attn_b1"], 1e-5)


@pytest.mark.parametrize("tag,use_lens,causal", [("mha_nomask", False, False), ("mha_kpm", True, False), ("mha_kpm_causal", True, True)])
def test_fused_attention_kernel_vs_reference(nn_, golden, tag, use_lens, causal, mfma_fp32):
    """return_attn_weights=False routes through the fused HIP kernel (Dh = 36 here: padded head dim path)."""
    g, x, lens, probe = block_inputs(golden)
    m = load_det_weights(nn_.RelPosMHAXL(144, 4, dropout=0.0, mask_pos_future=causal), "blk.mha.").to(DEV)
    pe = nn_.RelPosEncXL(144).to(DEV)(x)
    out = m(x, x, x, pe, key_lens=lens if use_lens else None, causal=causal, return_attn_weights=False)
    (out * probe).sum().backward()
    # bf16 MFMA operands (q+u, k, p, v, probabilities): 2^-8 relative per operand on |out| ~ 1
    close(out, g[f"{tag}:out"], 3e-2, 2e-2)
    rel = float((out.detach().cpu() - T(g[f"{tag}:out"])).norm() / T(g[f"{tag}:out"]).norm())
    assert rel < 8e-3, rel
    rel_dx = float((x.grad.cpu() - T(g[f"{tag}:dx"])).norm() / T(g[f"{tag}:dx"]).norm())
    assert rel_dx < 2e-2, rel_dx


@pytest.mark.parametrize("B,Tn,H,Dh", [(3, 250, 4, 64), (2, 125, 4, 64), (1, 333, 2, 64), (2, 40, 4, 36)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_fused_attention_full_width(ops, B, Tn, H, Dh, dtype, causal, mfma_fp32):
    """HIP kernel vs the fp32 formula (oracle/tsasr_ref.relpos_mha's core) on bf16-rounded operands."""
    D = H * Dh
    g = torch.Generator().manual_seed(Tn + Dh)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(dtype)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(dtype)
    u, v = torch.randn(Dh, H, generator=g) * 0.3, torch.randn(Dh, H, generator=g) * 0.3
    lens = torch.tensor([Tn, max(1, Tn // 2), max(1, Tn - 7)][:B], dtype=torch.int32)
    scale = 1.0 / D ** 0.5
    out, _ = ops.relpos_attention(qkv.to(DEV), pk.to(DEV), u.to(DEV), v.to(DEV), lens.to(DEV), H, scale, causal, 0.0, False)
    bfr = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    q, k, vv = qkv.float().view(B, Tn, H, 3 * Dh).chunk(3, dim=-1)
    uu, vb = u.reshape(-1).view(1, 1, H, Dh), v.reshape(-1).view(1, 1, H, Dh)
    p = bfr(pk.float()).view(1, -1, H, Dh)
    ac = torch.matmul(bfr(q + uu).transpose(1, 2), bfr(k).permute(0, 2, 3, 1))
    bd = torch.matmul(bfr(q + vb).transpose(1, 2), p.permute(0, 2, 3, 1))
    idx = torch.arange(Tn)
    bd = torch.gather(bd, 3, (idx[None, :] - idx[:, None] + Tn - 1).expand(B, H, Tn, Tn))
    sc = (ac + bd) * scale
    if causal:
        sc = sc.masked_fill(idx[None, :] > idx[:, None], float("-inf"))
    sc = sc.masked_fill((idx[None, :] >= lens[:, None]).view(B, 1, 1, Tn), float("-inf"))
    ref = torch.matmul(torch.softmax(sc, -1), bfr(vv).transpose(1, 2)).transpose(1, 2).reshape(B, Tn, D)
    rel = float((out.float().cpu() - ref).norm() / ref.norm())
    assert rel < (6e-3 if dtype == torch.float32 else 1e-2), rel   # probabilities are rounded to bf16 for P.V
    close(out, ref, 4e-2, 4e-2)


@pytest.mark.parametrize("B,Tn,H,Dh", [(3, 250, 4, 64), (2, 70, 4, 36), (1, 200, 2, 64), (1, 1100, 2, 36), (2, 1500, 1, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_fused_attention_backward_full_width(ops, B, Tn, H, Dh, dtype, causal, mfma_fp32):
    """HIP backward (dQ/dK/dV, d pos_bias_u/v, d pk) vs autograd through the ORACLE's attention (oracle/tsasr_ref.py relpos_core, the
    function the reference-generated golden vectors pin through relpos_mha) - not through product code. The two T > 1000 shapes take
    the long-sequence paths: keys split over workgroups in the forward and in the query-major backward (fp32 dQ shares + merge), query
    ranges split in the d(pk) pass."""
    D = H * Dh
    g = torch.Generator().manual_seed(Tn * 3 + Dh)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(dtype)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(dtype)
    u, v = torch.randn(Dh, H, generator=g) * 0.3, torch.randn(Dh, H, generator=g) * 0.3
    dout = torch.randn(B, Tn, D, generator=g).to(dtype)
    lens = torch.tensor([Tn, max(1, Tn // 2), max(1, Tn - 7)][:B], dtype=torch.int32)
    scale = 1.0 / D ** 0.5
    leaf = [t.float().clone().requires_grad_() for t in (qkv, pk, u, v)]
    R = importlib.import_module("oracle.tsasr_ref")
    pad = torch.arange(Tn)[None, :] >= lens[:, None].long()
    ref, _ = R.relpos_core(leaf[0], leaf[1], leaf[2], leaf[3], H, scale, pad, causal)
    ref.backward(dout.float())
    dev = [t.to(DEV).requires_grad_() for t in (qkv, pk, u, v)]
    out, _ = ops.relpos_attention(dev[0], dev[1], dev[2], dev[3], lens.to(DEV), H, scale, causal, 0.0, False)
    out.backward(dout.to(DEV))
    for a, r_, name in zip(dev, leaf, ("dqkv", "dpk", "du", "dv")):
        rel = float((a.grad.float().cpu() - r_.grad).norm() / r_.grad.norm())
        assert rel < (1.5e-2 if dtype == torch.float32 else 2.5e-2), (name, rel)   # bf16 MFMA operands incl. P and dS


@pytest.mark.parametrize("B,Tn,H,Dh", [(3, 250, 4, 64), (2, 70, 4, 36), (1, 333, 2, 64), (2, 40, 1, 8)])
@pytest.mark.parametrize("causal", [0, 1, 16])
def test_attention_exact_fp32_vs_oracle(ops, B, Tn, H, Dh, causal):
    """The fp32 parity path (csrc/attention_f32.hip: no matrix cores, no operand rounding) against autograd through the ORACLE's
    attention (oracle/tsasr_ref.relpos_core, pinned by the reference-generated golden vectors) on UN-rounded fp32 inputs: output and
    every gradient (dqkv, d pk, d pos_bias_u / v) to 2e-5 relative L2 - three orders tighter than the bf16-operand kernels allow. Ragged
    key lengths, look-ahead mask, the block-causal extension (C = 16), a padded head dim (36) and a tiny one (8)."""
    D = H * Dh
    g = torch.Generator().manual_seed(Tn * 5 + Dh + causal)
    qkv = torch.randn(B, Tn, 3 * D, generator=g)
    pk = torch.randn(2 * Tn - 1, D, generator=g)
    u, v = torch.randn(Dh, H, generator=g) * 0.3, torch.randn(Dh, H, generator=g) * 0.3
    dout = torch.randn(B, Tn, D, generator=g)
    lens = torch.tensor([Tn, max(1, Tn // 2), max(1, Tn - 7)][:B], dtype=torch.int32)
    scale = 1.0 / D ** 0.5
    leaf = [t.double().clone().requires_grad_() for t in (qkv, pk, u, v)]
    R = importlib.import_module("oracle.tsasr_ref")
    pad = torch.arange(Tn)[None, :] >= lens[:, None].long()
    ref, _ = R.relpos_core(leaf[0], leaf[1], leaf[2], leaf[3], H, scale, pad, causal)
    ref.backward(dout.double())
    dev = [t.to(DEV).requires_grad_() for t in (qkv, pk, u, v)]
    assert ops.ATTN_F32_EXACT
    out, _ = ops.relpos_attention(dev[0], dev[1], dev[2], dev[3], lens.to(DEV), H, scale, causal, 0.0, False)
    out.backward(dout.to(DEV))
    rel = float((out.detach().double().cpu() - ref.detach()).norm() / ref.detach().norm())
    assert rel < 2e-6, ("out", rel)
    for a, r_, name in zip(dev, leaf, ("dqkv", "dpk", "du", "dv")):
        rel = float((a.grad.double().cpu() - r_.grad).norm() / r_.grad.norm())
        assert rel < 2e-5, (name, rel)


def test_attention_exact_fp32_dropout_and_cross_shapes(ops):
    """(1) dropout: the backward regenerates the forward's mask - out is linear in V for a fixed mask, so <dout, out(V + dV) - out(V)> must
    equal <grad_V, dV>; the kept fraction is 1 - p to sampling error. (2) cross-attention shapes (Tq != Tk, no positions, separate q and
    kv tensors: the `cross_attention` speaker injection, models/conformer.py:263-266) against torch's formula in float64, values and
    gradients, with ragged key lengths."""
    B, Tn, H, Dh = 2, 96, 2, 64
    D = H * Dh
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(DEV)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(DEV)
    u, v = (torch.randn(Dh, H, generator=g) * 0.3).to(DEV), (torch.randn(Dh, H, generator=g) * 0.3).to(DEV)
    dout = torch.randn(B, Tn, D, generator=g).to(DEV)

    def run(x, seed):
        x4 = x.view(x.shape[0], x.shape[1], H, 3 * Dh)
        return ops._AttnF32Fn.apply(x4[..., :Dh], x4[..., Dh:2 * Dh], x4[..., 2 * Dh:], pk, u, v, None, H, 1.0 / D ** 0.5, 0, 0.3, seed)
    x = qkv.clone().requires_grad_()
    out = run(x, 1234)
    out.backward(dout)
    dV = torch.zeros_like(qkv)
    dV.view(B, Tn, H, 3 * Dh)[..., 2 * Dh:] = torch.randn(B, Tn, H, Dh, generator=torch.Generator().manual_seed(5)).to(DEV)
    out2 = run(qkv + dV, 1234)
    lhs = float(((out2 - out.detach()).double() * dout.double()).sum())
    rhs = float((x.grad.double() * dV.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    ones = torch.zeros_like(qkv)
    ones.view(B, Tn, H, 3 * Dh)[..., 2 * Dh:] = 1.0           # V = 1: out = sum of the dropped probabilities = kept fraction / (1 - p) on average
    kept = float(run(ones, 77).mean())
    assert abs(kept - 1.0) < 0.05, kept
    # cross-attention
    Tq, Tk = 50, 31
    q = torch.randn(B, Tq, D, generator=g)
    kv = torch.randn(B, Tk, 2 * D, generator=g)
    do = torch.randn(B, Tq, D, generator=g)
    lens = torch.tensor([Tk, 17], dtype=torch.int32)
    qd, kvd = q.double().requires_grad_(), kv.double().requires_grad_()
    qh = qd.view(B, Tq, H, Dh).transpose(1, 2)
    kh, vh = kvd[..., :D].view(B, Tk, H, Dh).transpose(1, 2), kvd[..., D:].view(B, Tk, H, Dh).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) / Dh ** 0.5
    sc = sc.masked_fill((torch.arange(Tk)[None, :] >= lens[:, None].long()).view(B, 1, 1, Tk), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, Tq, D)
    ref.backward(do.double())
    qg, kvg = q.to(DEV).requires_grad_(), kv.to(DEV).requires_grad_()
    kv5 = kvg.view(B, Tk, 2, H, Dh)
    o = ops.attention_f32(qg.view(B, Tq, H, Dh), kv5[:, :, 0], kv5[:, :, 1], H, 1.0 / Dh ** 0.5, lens.to(DEV), False, 0.0)
    o.backward(do.to(DEV))
    for got, want, name in ((o, ref, "out"), (qg.grad, qd.grad, "dq"), (kvg.grad, kvd.grad, "dkv")):
        rel = float((got.detach().double().cpu() - want.detach()).norm() / want.detach().norm())
        assert rel < 2e-5, (name, rel)


@pytest.mark.parametrize("B,U,I,H", [(4, 21, 28, 64), (2, 7, 28, 512), (3, 1, 5, 20)])
def test_lstm_exact_fp32_vs_float64_and_stepwise(ops, B, U, I, H):
    """The fp32 LSTM kernels (csrc/lstm_f32.hip: parity-mode predictor and the searchers' step-wise calls; torch.nn.LSTM behind
    SB/nnet/RNN.py:170-278, stepped by SB/decoders/transducer.py:246-353) against torch.nn.LSTM evaluated in float64 on the CPU: outputs,
    final state and every gradient (input, both weights, both biases, initial state) to 1e-5 relative L2; U single steps with a carried
    (h, c) reproduce the sequence call BIT FOR BIT (the same arithmetic in the same order)."""
    g = torch.Generator().manual_seed(U * 7 + H)
    rnn = torch.nn.LSTM(I, H, batch_first=True)
    x = torch.randn(B, U, I, generator=g)
    h0, c0 = torch.randn(1, B, H, generator=g) * 0.5, torch.randn(1, B, H, generator=g) * 0.5
    dout, dhn = torch.randn(B, U, H, generator=g), torch.randn(1, B, H, generator=g)
    ref = torch.nn.LSTM(I, H, batch_first=True).double()
    ref.load_state_dict({k: v.double() for k, v in rnn.state_dict().items()})
    xr, h0r, c0r = x.double().requires_grad_(), h0.double().requires_grad_(), c0.double().requires_grad_()
    out_r, (hn_r, cn_r) = ref(xr, (h0r, c0r))
    ((out_r * dout.double()).sum() + (hn_r * dhn.double()).sum() + cn_r.sum()).backward()
    dev = rnn.to(DEV)
    xg, h0g, c0g = x.to(DEV).requires_grad_(), h0.to(DEV).requires_grad_(), c0.to(DEV).requires_grad_()
    assert ops.LSTM_F32_HIP and ops.lstm_f32_ok(dev)
    out, (hn, cn) = ops.lstm(xg, dev, (h0g, c0g))
    ((out * dout.to(DEV)).sum() + (hn * dhn.to(DEV)).sum() + cn.sum()).backward()
    rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).norm() / (b.detach().norm() + 1e-30))  # noqa: E731
    for a, b, name in ((out, out_r, "out"), (hn, hn_r, "hn"), (cn, cn_r, "cn"), (xg.grad, xr.grad, "dx"), (h0g.grad, h0r.grad, "dh0"),
                       (c0g.grad, c0r.grad, "dc0"), (dev.weight_ih_l0.grad, ref.weight_ih_l0.grad, "dW_ih"),
                       (dev.weight_hh_l0.grad, ref.weight_hh_l0.grad, "dW_hh"), (dev.bias_ih_l0.grad, ref.bias_ih_l0.grad, "db_ih"),
                       (dev.bias_hh_l0.grad, ref.bias_hh_l0.grad, "db_hh")):
        assert rel(a, b) < 1e-5, (name, rel(a, b))
    with torch.no_grad():
        hid, outs = (h0g.detach(), c0g.detach()), []
        for t in range(U):
            o, hid = ops.lstm(x[:, t:t + 1].to(DEV), dev, hid)
            outs.append(o)
        assert torch.equal(torch.cat(outs, 1), out.detach()) and torch.equal(hid[0], hn.detach()) and torch.equal(hid[1], cn.detach())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["sum", "prod"])
def test_speaker_injection_kernels(ops, dtype, mode):
    """tsasr_inject_fwd / _bwd (the `sum` / `prod` injections, models/conformer.py:247-253) against the broadcast expression in float64:
    values, dsrc and the time-summed dspk; odd T, D not a multiple of 64."""
    B, Tn, D = 3, 37, 72
    g = torch.Generator().manual_seed(3)
    src, spk, dout = torch.randn(B, Tn, D, generator=g), torch.randn(B, 1, D, generator=g), torch.randn(B, Tn, D, generator=g)
    a, b = src.to(dtype).double().requires_grad_(), spk.to(dtype).double().requires_grad_()
    ref = a * b if mode == "prod" else a + b
    ref.backward(dout.to(dtype).double())
    x, y = src.to(DEV).to(dtype).requires_grad_(), spk.to(DEV).to(dtype).requires_grad_()
    assert ops.inject_ok(x, y)
    out = ops.inject(x, y, mode)
    out.backward(dout.to(DEV).to(dtype))
    tol = 1e-6 if dtype == torch.float32 else 6e-3     # bf16: one rounding of each output (and of the 37-term sum)
    for got, want in ((out, ref), (x.grad, a.grad), (y.grad, b.grad)):
        rel = float((got.detach().double().cpu() - want.detach()).norm() / want.detach().norm())
        assert rel < tol, (mode, rel)


def test_fused_attention_dropout_consistency(ops):
    """With dropout the backward must regenerate the forward's mask: check d(out)/d(V) . dout == out . dout structure via
    finite differences on V (out is linear in V for a fixed mask)."""
    B, Tn, H, Dh = 2, 96, 2, 64
    D = H * Dh
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(DEV)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(DEV)
    u, v = (torch.randn(Dh, H, generator=g) * 0.3).to(DEV), (torch.randn(Dh, H, generator=g) * 0.3).to(DEV)
    dout = torch.randn(B, Tn, D, generator=g).to(DEV)
    fn = ops._RelPosAttnFn
    x = qkv.clone().requires_grad_()
    out = fn.apply(x, pk, u, v, None, H, 1.0 / D ** 0.5, False, 0.3, 1234)
    out.backward(dout)
    # out is linear in V:  <dout, out(V + dV) - out(V)> == <grad_V, dV>
    dV = torch.zeros_like(qkv)
    dV.view(B, Tn, H, 3 * Dh)[..., 2 * Dh:] = torch.randn(B, Tn, H, Dh, generator=torch.Generator().manual_seed(5)).to(DEV)
    out2 = fn.apply(qkv + dV, pk, u, v, None, H, 1.0 / D ** 0.5, False, 0.3, 1234)
    lhs = float(((out2 - out.detach()).double() * dout.double()).sum())
    rhs = float((x.grad.double() * dV.double()).sum())
    assert lhs == pytest.approx(rhs, rel=3e-2), (lhs, rhs)
    # and a different seed gives a different mask
    out3 = fn.apply(qkv, pk, u, v, None, H, 1.0 / D ** 0.5, False, 0.3, 99)
    assert float((out3 - out.detach()).abs().max()) > 1e-2


@pytest.mark.parametrize("Tn,p", [(250, 0.1), (125, 0.5)])
def test_attention_dropout_stream_statistics(ops, Tn, p):
    """The attention dropout stream (csrc/attn_common.h: one 24-bit-multiply word per two keys on a strong per-row hash). The keep-bits the
    short-sequence forward hands to the backward ARE the mask: they equal the numpy twin (tests/helpers/attn_mask.py) bit for bit, the
    keep rate is 1 - p to sampling error (p quantised to 1/65536), and neighbours along a row, along a column, along both diagonals and
    32 keys apart (same position of consecutive blocks) are uncorrelated."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import attn_mask
    B, H, Dh = 8, 4, 64
    D = H * Dh
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(DEV, torch.bfloat16)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(DEV, torch.bfloat16)
    u, v = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    L = ops.C.lib()
    kb = torch.zeros(L.tsasr_relpos_attn_keepbits_bytes(B, Tn, H), dtype=torch.uint8, device=DEV)
    out, lse = torch.empty(B, Tn, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, Tn, device=DEV)
    seed = 0x1234567
    L.tsasr_relpos_attn_keepbits(ops.C.ptr(kb))
    assert L.tsasr_relpos_attn_fwd(ops.C.ptr(qkv), ops.C.ptr(pk), ops.C.ptr(u), ops.C.ptr(v), None, ops.C.ptr(out), ops.C.ptr(lse), B, Tn, H, Dh, 1.0 / D ** 0.5,
                                   0, p, seed, None, ops.C.BF16, ops.C.stream_ptr()) == 0
    torch.cuda.synchronize()
    bits = kb.view(torch.int16).cpu().numpy().view(np.uint16).reshape(B * H * Tn, 2, 8)        # [row][hh][block of 32 keys]
    j = np.arange(Tn)
    jl = j & 31
    hh, gidx = (jl >> 2) & 1, (jl & 3) + 4 * (jl >> 3)                                         # accumulator element g of key jl: (g&3) + 8(g>>2) + 4hh
    got = ((bits[:, hh, j >> 5] >> gidx[None, :].astype(np.uint16)) & 1).astype(bool).reshape(B, H, Tn, Tn)
    ref = attn_mask.keep_mask(B, H, Tn, p, seed)
    assert np.array_equal(got, ref)
    m = got.reshape(B * H * Tn, Tn).astype(np.float64)
    n = m.size
    sig = (p * (1 - p) / n) ** 0.5
    assert abs(m.mean() - (1 - p)) < 5 * sig + 1e-5

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).mean() / (a.std() * b.std()))

    lim = 6 / n ** 0.5
    for d in (1, 2, 4, 8, 16, 32):
        assert abs(corr(m[:, :-d], m[:, d:])) < lim, d
    mm = got.reshape(B * H, Tn, Tn).astype(np.float64)
    assert abs(corr(mm[:, :-1], mm[:, 1:])) < lim and abs(corr(mm[:, :-1, :-1], mm[:, 1:, 1:])) < lim and abs(corr(mm[:, :-1, 1:], mm[:, 1:, :-1])) < lim
    # and the mask does what a mask does: with V = 1 the output is the kept probability mass / (1 - p)
    qkv1 = qkv.clone()
    qkv1.view(B, Tn, H, 3 * Dh)[..., 2 * Dh:] = 1.0
    L.tsasr_relpos_attn_fwd(ops.C.ptr(qkv1), ops.C.ptr(pk), ops.C.ptr(u), ops.C.ptr(v), None, ops.C.ptr(out), ops.C.ptr(lse), B, Tn, H, Dh, 1.0 / D ** 0.5,
                            0, p, seed, None, ops.C.BF16, ops.C.stream_ptr())
    assert abs(float(out.float().mean()) - 1.0) < 0.02


@pytest.mark.parametrize("Tn,causal,ragged", [(250, False, True), (96, True, False), (33, False, True), (256, False, False)])
def test_attention_keepbits_from_the_forward_give_the_same_gradients(ops, Tn, causal, ragged):
    """Short bf16 sequences: the forward stores the dropout keep-bits it hashed and the backward reads them
    (tsasr_relpos_attn_keepbits). Every gradient must equal, bit for bit, the one of a backward that hashes the mask again."""
    B, H, Dh = 3, 4, 64
    D = H * Dh
    g = torch.Generator().manual_seed(17)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(DEV, torch.bfloat16)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(DEV, torch.bfloat16)
    u, v = (torch.randn(Dh, H, generator=g) * 0.3).to(DEV), (torch.randn(Dh, H, generator=g) * 0.3).to(DEV)
    dout = torch.randn(B, Tn, D, generator=g).to(DEV, torch.bfloat16)
    lens = torch.tensor([Tn, max(1, Tn - 37), max(1, Tn // 2 + 1)], dtype=torch.int32, device=DEV) if ragged else None
    fn = ops._RelPosAttnFn
    assert ops.C.lib().tsasr_relpos_attn_keepbits_bytes(B, Tn, H) == B * H * Tn * 32
    grads = {}
    prev = ops._ATTN_KEEPBITS
    try:
        for on in (True, False):
            ops._ATTN_KEEPBITS = on
            leaves = [t.clone().requires_grad_() for t in (qkv, pk, u, v)]
            out = fn.apply(leaves[0], leaves[1], leaves[2], leaves[3], lens, H, 1.0 / D ** 0.5, causal, 0.1, 4321)
            assert (out.grad_fn.keepbits is not None) == on
            out.backward(dout)
            grads[on] = [out.detach()] + [t.grad for t in leaves]
    finally:
        ops._ATTN_KEEPBITS = prev
    for a, b_, name in zip(grads[True], grads[False], ("out", "dqkv", "dpk", "du", "dv")):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a, b_), name


def test_fused_clip_adamw_vs_torch():
    """csrc/optim.hip vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (the reference's optimizer step)."""
    opt_mod = importlib.import_module("ts-asr_amd.optim")
    n = 1_000_003  # not a multiple of 4: tail path
    g = torch.Generator().manual_seed(3)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.AdamW([p], lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    norm = torch.zeros((), device=DEV)
    for t in range(1, 4):
        p.grad = gr.clone() * t
        total = torch.nn.utils.clip_grad_norm_([p], 5.0)
        ref.step()
        opt_mod._clip_adamw(pd, (gr * t).to(DEV), m, v, norm, 1e-3, 0.9, 0.98, 1e-8, 0.01, t, 5.0)
        assert float(norm) == pytest.approx(float(gr.double().norm()) * t, rel=2e-6)   # fp64 yardstick (torch's fp32 CPU sum is off by 1e-5)
        assert float(norm) == pytest.approx(float(total), rel=1e-4)
    np.testing.assert_allclose(pd.cpu().numpy(), p.detach().numpy(), atol=2e-6, rtol=1e-5)


def test_fused_clip_adamw_skips_a_nonfinite_gradient():
    """`skip_nonfinite_step: True` (a counter is passed): a NaN / Inf in the gradient makes the global norm non-finite, the kernel leaves p, m,
    v (and the bf16 shadow) untouched and counts the skipped step in its own device float; a finite gradient afterwards updates normally and
    leaves the counter alone. Without a counter (the default, `skip_nonfinite_step: False`) the step is applied as the reference applies it
    (SB/core.py:1072-1093): torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the same buffers give the same (poisoned) weights."""
    opt_mod = importlib.import_module("ts-asr_amd.optim")
    n = 4099
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    pd, m, v = p0.to(DEV), torch.full((n,), 0.25, device=DEV), torch.full((n,), 0.5, device=DEV)
    p16 = pd.to(torch.bfloat16)
    norm, skipped = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    guard = torch.full((64,), 7.0, device=DEV)      # a 1-float norm buffer must not be written past its end (round-2 advisor finding)
    norm1 = guard[31:32]
    for bad in (float("nan"), float("inf")):
        gr = torch.randn(n, generator=g)
        gr[17] = bad
        opt_mod._clip_adamw(pd, gr.to(DEV), m, v, norm1, 1e-3, 0.9, 0.98, 1e-8, 0.01, 1, 5.0, p16=p16, skipped_out=skipped)
        assert torch.equal(pd.cpu(), p0) and float(m.min()) == 0.25 == float(m.max()) and float(v.min()) == 0.5 == float(v.max())
        assert torch.equal(p16.cpu(), p0.to(torch.bfloat16))
        assert not np.isfinite(float(norm1))
    assert float(skipped) == 2.0
    assert torch.equal(torch.cat([guard[:31], guard[32:]]), torch.full((63,), 7.0, device=DEV))
    opt_mod._clip_adamw(pd, torch.randn(n, generator=g).to(DEV), m, v, norm, 1e-3, 0.9, 0.98, 1e-8, 0.01, 1, 5.0, p16=p16, skipped_out=skipped)
    assert float(skipped) == 2.0 and np.isfinite(float(norm)) and not torch.equal(pd.cpu(), p0)
    # reference behaviour (no counter): the update goes ahead; compare with torch on the same numbers
    for bad in (float("nan"), float("inf")):
        pr = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
        gr = torch.randn(n, generator=g)
        gr[17] = bad
        pr.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([pr], 5.0)
        opt.step()
        pd2, m2, v2 = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        opt_mod._clip_adamw(pd2, gr.to(DEV), m2, v2, None, 1e-3, 0.9, 0.98, 1e-8, 0.01, 1, 5.0)   # both outputs optional
        ref_nan, got_nan = torch.isnan(pr.detach()), torch.isnan(pd2.cpu())
        assert torch.equal(ref_nan, got_nan) and bool(got_nan.any())                               # the same elements are poisoned
        np.testing.assert_allclose(pd2.cpu()[~got_nan].numpy(), pr.detach()[~ref_nan].numpy(), atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(8000, 2048, 256), (8000, 256, 2048), (250, 144, 144), (1000, 640, 256), (129, 72, 200), (2048, 256, 8000),
                                   (4000, 256, 2048), (1000, 200, 1024)])   # the last two: 64x64 tiles at long K (wave-K main loops), ragged M / N
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bf16_layouts(ops, M, N, K, ta, tb):
    """csrc/gemm.hip vs fp32 matmul of the same bf16 operands: every operand layout, ragged edges, split-K accumulate."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    A = (a.t().contiguous() if ta else a).to(DEV)      # [K,M] when transA
    Bm = (b.t().contiguous() if tb else b).to(DEV)     # [K,N] when transB
    if (M % 8 and ta) or (N % 8 and tb):
        pytest.skip("transposed operands need 8-element rows")
    lda, ldb = (M if ta else K), (N if tb else K)
    out = ops.gemm_bf16(A, Bm, M, N, K, lda, ldb, ta, tb)
    rel = float((out.float().cpu() - ref).norm() / ref.norm())
    assert rel < 4e-3, rel                               # bf16 output rounding
    acc = torch.ones(M, N, device=DEV)
    ops.gemm_bf16(A, Bm, M, N, K, lda, ldb, ta, tb, out=acc, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy(), (ref + 1).numpy(), atol=2e-3 * K ** 0.5, rtol=1e-4)
    f32 = ops.gemm_bf16(A, Bm, M, N, K, lda, ldb, ta, tb, out_dtype=torch.float32)
    np.testing.assert_allclose(f32.cpu().numpy(), ref.numpy(), atol=2e-3 * K ** 0.5, rtol=1e-4)


def test_linear_fn_grad_sink(ops):
    """_LinearFn: forward/dgrad/wgrad vs torch, and the weight gradient lands in a registered arena without autograd's help."""
    dp = importlib.import_module("ts-asr_amd.dp")
    lin = torch.nn.Linear(256, 512, bias=False).to(DEV)
    mods = torch.nn.ModuleDict({"l": lin})
    arena = dp.GradArena(mods)
    ops.set_grad_sink(arena)
    try:
        x = torch.randn(4, 50, 256, device=DEV).to(torch.bfloat16).requires_grad_()
        arena.begin_backward(False)
        y = ops.matmul_nt(x, lin.weight)
        dy = torch.randn_like(y)
        y.backward(dy)
        arena.finish_backward()
        w16 = lin.weight.detach().to(torch.bfloat16).float()
        close(y, x.detach().float() @ w16.t(), 3e-2, 2e-2)
        close(x.grad, dy.float() @ w16, 6e-2, 2e-2)
        ref_dw = dy.float().reshape(-1, 512).t() @ x.detach().float().reshape(-1, 256)
        close(lin.weight.grad, ref_dw, 2e-2, 1e-3)
        assert lin.weight.grad.data_ptr() == arena.grads[arena.offset[id(lin.weight)]:].data_ptr()
    finally:
        ops.set_grad_sink(None)


@pytest.mark.parametrize("collect", [True, False])
def test_linear_cols_weight_gradient_lands_in_its_columns(ops, collect):
    """The `cat` injection's projection as two column ranges of ONE weight (models/conformer.py:254-262): both halves' weight gradients
    go into their own columns of the weight's arena slot - through the grouped launch (collect_wgrads on) and through the split-K GEMM
    fallback, whose output is a strided view (row stride 2D, not D) - against the fp32 product of the same bf16 operands."""
    dp = importlib.import_module("ts-asr_amd.dp")
    D = 256
    lin = torch.nn.Linear(2 * D, D, bias=True).to(DEV)
    guard = torch.nn.Linear(D, 8, bias=False).to(DEV)          # the arena slot right behind the weight: must stay untouched
    mods = torch.nn.ModuleDict({"l": lin, "g": guard})
    arena = dp.GradArena(mods)
    arena.collect_wgrads = collect
    ops.set_grad_sink(arena)
    try:
        g = torch.Generator().manual_seed(11)
        src = torch.randn(4, 50, D, generator=g).to(DEV).to(torch.bfloat16).requires_grad_()
        spk = torch.randn(4, 1, D, generator=g).to(DEV).to(torch.bfloat16).requires_grad_()
        assert ops.linear_cols_ok(src, lin.weight, 0, D) and ops.linear_cols_ok(spk, lin.weight, D, D)
        arena.begin_backward(False)
        y = ops.linear_cols(src, lin.weight, None, 0, D) + ops.linear_cols(spk, lin.weight, lin.bias, D, D)
        dy = torch.randn(y.shape, generator=g).to(DEV).to(torch.bfloat16)
        y.backward(dy)
        arena.finish_backward()
        torch.cuda.synchronize()
        w16 = lin.weight.detach().to(torch.bfloat16).float()
        ref = src.detach().float() @ w16[:, :D].t() + spk.detach().float() @ w16[:, D:].t() + lin.bias.detach().float()
        close(y, ref, 6e-2, 2e-2)
        dyf = dy.float()
        ref_dw = torch.cat([dyf.reshape(-1, D).t() @ src.detach().float().reshape(-1, D),
                            dy.sum(1).float().t() @ spk.detach().float().reshape(-1, D)], dim=1)   # autograd's broadcast sum is rounded to bf16 once
        close(lin.weight.grad, ref_dw, 2e-2, 2e-3)
        assert float(guard.weight.grad.abs().max()) == 0.0
        close(src.grad, dyf @ w16[:, :D], 6e-2, 2e-2)
    finally:
        ops.set_grad_sink(None)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 29, 5), (64, 64, 16), (130, 257, 100), (500, 96, 2048)])
def test_gemm_f32_layouts_vs_float64(ops, M, N, K):
    """The parity mode's fp32 GEMM (csrc/gemm_f32.hip, fp32 matrix cores): the four operand layouts, padded leading dimensions,
    accumulate; against the product in float64. Tolerance: fp32 sums of K terms (no operand rounding): 2e-6 * sqrt(K) relative."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    ref = (a.double() @ b.double().t())
    tol = 2e-6 * max(K, 1) ** 0.5 * float(ref.abs().max() + 1.0)
    for ta in (0, 1):
        for tb in (0, 1):
            A = (a.t().contiguous() if ta else a)
            Bm = (b.t().contiguous() if tb else b)
            lda, ldb = A.shape[1] + 3, Bm.shape[1] + 5            # rows padded: strides differ from the row lengths
            Ap = torch.zeros(A.shape[0], lda).copy_(torch.nn.functional.pad(A, (0, 3))).to(DEV)
            Bp = torch.zeros(Bm.shape[0], ldb).copy_(torch.nn.functional.pad(Bm, (0, 5))).to(DEV)
            out = ops.gemm_f32(Ap, Bp, M, N, K, lda, ldb, ta, tb)
            assert float((out.double().cpu() - ref).abs().max()) <= tol, (ta, tb)
            acc = torch.full((M, N + 2), 0.5, device=DEV)
            ops.gemm_f32(Ap, Bp, M, N, K, lda, ldb, ta, tb, out=acc[:, :N], accumulate=True)
            assert float((acc[:, :N].double().cpu() - 0.5 - ref).abs().max()) <= tol + 1e-6
            assert float((acc[:, N:] - 0.5).abs().max()) == 0.0      # nothing written beyond N


def test_fp32_linear_runs_on_the_hip_gemm_and_equals_the_library(ops, monkeypatch):
    """ops.matmul_nt / ops.linear with fp32 operands (the parity mode): forward, data and weight gradient on tsasr_gemm_f32; same
    numbers as the library GEMM within fp32 summation-order noise."""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 50, 96, generator=g).to(DEV).requires_grad_()
    w = torch.randn(40, 96, generator=g).to(DEV).requires_grad_()
    b = torch.randn(40, generator=g).to(DEV).requires_grad_()
    dy = torch.randn(3, 50, 40, generator=g).to(DEV)
    outs = []
    for hip in (True, False):
        monkeypatch.setattr(ops, "_F32_HIP_GEMM", hip)
        for t in (x, w, b):
            t.grad = None
        y = ops.linear(x, w, b)
        assert ("_LinearF32Fn" in type(y.grad_fn).__name__ or "Add" in type(y.grad_fn).__name__) == hip or not hip
        y.backward(dy)
        outs.append([y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()])
    for a_, b_ in zip(*outs):
        assert float((a_ - b_).abs().max()) <= 2e-5 * float(b_.abs().max() + 1.0)


@pytest.mark.parametrize("B,U,H", [(8, 21, 128), (8, 21, 256), (40, 9, 256), (32, 121, 512), (1, 160, 512)])
def test_lstm_hip_path_vs_oracle(ops, B, U, H):
    """bf16 predictor LSTM vs the oracle's explicit recurrence (oracle/tsasr_ref.lstm), fwd + bwd. H = 128 runs the per-step
    kernels, H in {256, 512} the persistent whole-sequence kernels (one and two 32-row batch groups, ragged last group; the
    last case is the predictor's shape in BASELINE configs[1])."""
    from oracle import tsasr_ref as R
    I = 28
    g = torch.Generator().manual_seed(9)
    rnn = torch.nn.LSTM(I, H, batch_first=True).to(DEV)
    tok = torch.randint(0, 29, (B, U), generator=g)
    x = R.one_hot_embedding(tok, 29, 0)
    dout = torch.randn(B, U, H, generator=g)
    sd = {"rnn." + k: v.detach().cpu().clone().requires_grad_() for k, v in rnn.state_dict().items()}
    ref, _ = R.lstm(x, sd, "")
    ref.backward(dout)
    xg = x.to(DEV).to(torch.bfloat16)
    out, _ = ops.lstm(xg, rnn)
    out.backward(dout.to(DEV).to(torch.bfloat16))
    assert float((out.float().cpu() - ref.detach()).norm() / ref.detach().norm()) < 1e-2
    for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        a, r_ = getattr(rnn, k).grad.cpu(), sd["rnn." + k].grad
        assert float((a - r_).norm() / r_.norm()) < 3e-2, k


@pytest.mark.parametrize("B,U,H,blank", [(8, 21, 256, 0), (32, 121, 512, 0), (5, 7, 128, 3)])
def test_onehot_predictor_vs_oracle(ops, B, U, H, blank):
    """embedding -> decoder in one call (nnet.LSTM.forward_tokens: a token selects a column of W_ih, tsasr_lstm_onehot_gates) against
    the oracle's one-hot embedding + explicit recurrence, forward, lengths mask and every parameter gradient; blank = 3 exercises the
    tokens below the blank index (SB/nnet/embedding.py:83-90)."""
    from importlib import import_module
    from oracle import tsasr_ref as R
    nnet = import_module("ts-asr_amd.nnet")
    V = 29
    g = torch.Generator().manual_seed(11)
    emb = nnet.Embedding(V, consider_as_one_hot=True, blank_id=blank).to(DEV)
    dec = nnet.LSTM(H, input_size=V - 1, re_init=True).to(DEV)
    tok = torch.randint(0, V, (B, U), generator=g)
    rel = torch.linspace(1.0, 0.5, B)
    dout = torch.randn(B, U, H, generator=g)
    sd = {"rnn." + k: v.detach().cpu().clone().requires_grad_() for k, v in dec.rnn.state_dict().items()}
    ref, _ = R.lstm(R.one_hot_embedding(tok, V, blank), sd, "")
    keep = (torch.arange(U)[None, :] < torch.floor(rel * U)[:, None]).float()[..., None]
    ref = ref * keep
    ref.backward(dout)
    assert ops.lstm_onehot_supported(tok.to(DEV), dec.rnn, V)
    nnet.set_compute_dtype(torch.bfloat16)      # (this module's fixtures run the stage tests in fp32)
    try:
        out, _ = dec.forward_tokens(tok.to(DEV), emb, lengths=rel.to(DEV))
        assert out.dtype == torch.bfloat16
        out.backward(dout.to(DEV).to(torch.bfloat16))
        grads = {k: getattr(dec.rnn, k).grad.clone() for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")}
        for p_ in dec.parameters():
            p_.grad = None
        out2, _ = dec(emb(tok.to(DEV)), lengths=rel.to(DEV))     # the two-module form (F.embedding + the GEMM input projection)
    finally:
        nnet.set_compute_dtype(torch.float32)
    assert float((out.float().cpu() - ref.detach()).norm() / ref.detach().norm()) < 1e-2
    assert float(out.float().cpu()[keep.expand_as(ref) == 0].abs().max()) == 0.0
    for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
        a, r_ = grads[k].cpu(), sd["rnn." + k].grad
        assert float((a - r_).norm() / r_.norm()) < 3e-2, k
    assert float((out2.float() - out.float()).norm() / out.float().norm()) < 1e-2      # same function, two routes


def test_lstm_persistent_kernels_survive_graph_replay_with_dirty_workspace():
    """The whole-sequence kernels rely on arrival counters in the first 256 workspace bytes being zero at launch. Inside a
    replayed hipGraph the workspace is recycled memory: fill it (counters AND exchange payload) with 0xFF between replays; every
    replay must give the bits of the eager call (a captured hipMemsetAsync node failed exactly this from the 2nd replay on)."""
    from importlib import import_module
    C = import_module("ts-asr_amd._capi")
    B, U, H = 32, 33, 512
    g = torch.Generator().manual_seed(3)
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
        C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nb,
                                       C.stream_ptr()), "bwd")

    run()
    torch.cuda.synchronize()
    h_ref, dg_ref = h.clone(), dgates.clone()
    assert torch.isfinite(h_ref.float()).all() and torch.isfinite(dg_ref.float()).all()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for _ in range(4):
        ws.fill_(0xFF)
        h.zero_()
        dgates.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(h, h_ref) and torch.equal(dgates, dg_ref)


@pytest.mark.parametrize("U", [1, 2, 97, 640])
def test_lstm_single_utterance_kernels_vs_exchange_groups(monkeypatch, U):
    """B = 1 (the long-form configuration): the wave-autonomous recurrence (csrc/lstm.hip lstm_seq1_*: payload published into the
    0xFFFF-pre-filled outputs, no counters, no barriers) against the exchange-group kernels on the same inputs. Forward: the same MFMA
    chain in the same order - h, c and the activated gates BIT FOR BIT. Backward: the reduction over 4H runs as two halves instead of
    four quarters (fp32), dgates to 2e-2 relative (bf16 values; a last-bit flip early in the recurrence is carried along). Then as a
    replayed hipGraph over outputs left dirty (zeros: NOT the fill pattern) by the caller."""
    from importlib import import_module
    C = import_module("ts-asr_amd._capi")
    B, H = 1, 512
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
        C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nb,
                                       C.stream_ptr()), "bwd")
        torch.cuda.synchronize()
        assert int(ws[:8].view(torch.int32)[1]) == 0      # the error word
        return h.clone(), c.clone(), gates.clone(), dgates.clone()

    monkeypatch.setenv("TSASR_LSTM_SEQ1", "0")
    h0, c0, g0, d0 = run()
    monkeypatch.setenv("TSASR_LSTM_SEQ1", "1")
    h1, c1, g1, d1 = run()
    assert torch.isfinite(h1.float()).all() and torch.isfinite(d1.float()).all()
    assert torch.equal(h1, h0) and torch.equal(c1, c0) and torch.equal(g1, g0)
    assert float((d1.float() - d0.float()).norm() / d0.float().norm()) < 2e-2
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gates.copy_(gates0)
        C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh), B, U, H, C.BF16, C.ptr(ws), nb, C.stream_ptr()), "fwd")
        C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16, C.ptr(ws), nb,
                                       C.stream_ptr()), "bwd")
    for _ in range(3):
        ws.fill_(0xFF)
        h.zero_()
        dgates.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(h, h1) and torch.equal(dgates, d1)


def test_fbank_and_sentence_norm_known_answers(nn_, ops):
    """Reference known answers (vendor/speechbrain/tests/unittests/test_features.py:57-113) on the HIP kernels + oracle parity."""
    from oracle import tsasr_ref as R
    fb = nn_.Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).to(DEV)
    # silence -> every bin at amin -> -100 dB (and the top_db floor leaves it there)
    z = fb(torch.zeros(2, 3200, device=DEV))
    assert z.shape == (2, 21, 80) and torch.all(z == -100.0)
    # random waveforms incl. a ragged length (L not a multiple of the hop) vs the oracle
    g = torch.Generator().manual_seed(4)
    wav = torch.randn(3, 16000 + 77, generator=g) * 0.1
    wav[1, 9000:] = 0
    out = fb(wav.to(DEV))
    ref = R.fbank(wav)
    assert out.shape == ref.shape
    close(out, ref, 3e-3, 1e-4)          # dB scale; fp32 FFT vs rocFFT/pocketfft summation order
    # InputNormalization known answer: [1,2,3,0,0,0] with relative length 0.5 -> [-1,0,1,-2,-2,-2]
    norm = nn_.InputNormalization(norm_type="sentence")
    x = torch.tensor([1.0, 2, 3, 0, 0, 0], device=DEV).view(1, 6, 1)
    assert torch.equal(norm(x, torch.tensor([0.5], device=DEV)).squeeze().cpu(), torch.tensor([-1.0, 0, 1, -2, -2, -2]))
    lens = torch.tensor([1.0, 0.6, 0.33])
    close(norm(out, lens.to(DEV)), R.sentence_norm(ref, lens), 2e-3, 1e-3)
    xb = torch.randn(2, 50, 144, generator=g)       # feature width that does not divide 256
    close(ops.sentence_norm(xb.to(DEV), torch.tensor([50, 20], device=DEV), 1e-10), R.sentence_norm(xb, torch.tensor([1.0, 0.4])), 1e-5)


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("M", [777, 3999, 7995])   # 3999 / 7995 rows: the 8-wave 128- and 256-row tiles of csrc/gemm_big.hip, ragged last tile
def test_fused_ffn_core(ops, p, M):
    """_FFNFn (two HIP GEMMs with fused bias/LeakyReLU/dropout epilogues) vs the unfused composition of already-tested ops
    sharing the same counter-based dropout stream, forward and every gradient."""
    D, F1 = 256, 2048
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, M // 3, D, generator=g).to(torch.bfloat16)
    w1, b1 = torch.randn(F1, D, generator=g) / D ** 0.5, torch.randn(F1, generator=g) * 0.1
    w2 = torch.randn(D, F1, generator=g) / F1 ** 0.5
    do = torch.randn(3, M // 3, D, generator=g).to(torch.bfloat16)
    outs = []
    for fused in (True, False):
        xs = x.to(DEV).requires_grad_()
        ps = [t.to(DEV).requires_grad_() for t in (w1, b1, w2)]
        ops._seed_counter[0] = 100            # same per-call dropout stream id for both variants
        if fused:
            o = ops.ffn_core(xs, ps[0], ps[1], ps[2], 0.01, p, True)
        else:
            hmid = ops.bias_act_dropout(ops.matmul_nt(xs, ps[0]), ps[1], 0.01, p, True)
            o = ops.matmul_nt(hmid, ps[2])
        o.backward(do.to(DEV))
        outs.append([o.detach().float().cpu(), xs.grad.float().cpu()] + [t.grad.float().cpu() for t in ps])
    for a, b_, name in zip(outs[0], outs[1], ("out", "dx", "dw1", "db1", "dw2")):
        rel = float((a - b_).norm() / b_.norm())
        assert rel < (6e-3 if name == "out" else 1.5e-2), (name, rel)   # the fused epilogues round to bf16 once instead of twice
    if p == 0.0:  # and against plain fp32 math
        ref = torch.nn.functional.leaky_relu(x.float() @ w1.to(torch.bfloat16).float().t() + b1, 0.01).to(torch.bfloat16).float() @ w2.to(torch.bfloat16).float().t()
        assert float((outs[0][0] - ref).norm() / ref.norm()) < 6e-3


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("M", [3999, 7995])
def test_ffn_mask_words_give_the_same_bits(ops, monkeypatch, p, M):
    """The FFN's data gradient reading the forward epilogue's mask words (keep-bits + sign of the stored activation, uint16 per 8
    outputs) instead of the saved activation and a re-hash of the keep-bits: every output and gradient bit-identical to the form without
    them. The words need the arena's transposed weight shadow (the k-contiguous backward GEMM): provided by hand here."""
    D, F1 = 256, 2048
    g = torch.Generator().manual_seed(22)
    x = torch.randn(M, D, generator=g).to(torch.bfloat16)
    w1, b1 = torch.randn(F1, D, generator=g) / D ** 0.5, torch.randn(F1, generator=g) * 0.1
    w2 = torch.randn(D, F1, generator=g) / F1 ** 0.5
    do = torch.randn(M, D, generator=g).to(torch.bfloat16)
    outs = []
    for masks in (True, False):
        monkeypatch.setattr(ops, "_FFN_MASK", masks)
        xs = x.to(DEV).requires_grad_()
        ps = [t.to(DEV).requires_grad_() for t in (w1, b1, w2)]
        ps[2]._bf16 = ps[2].detach().to(torch.bfloat16)
        ps[2]._bf16_t = ps[2]._bf16.t().contiguous()
        ps[2]._bf16_ver = ps[2]._version
        assert ops.fused_mask_ok(M, F1, D) and ops._bf16_weight_t(ps[2]) is not None
        ops._seed_counter[0] = 100
        o = ops.ffn_core(xs, ps[0], ps[1], ps[2], 0.01, p, True)
        o.backward(do.to(DEV))
        outs.append([o.detach(), xs.grad] + [t.grad for t in ps])
    for a, b_, name in zip(outs[0], outs[1], ("out", "dx", "dw1", "db1", "dw2")):
        assert torch.equal(a, b_), name


# ---------------------------------------------------------------------------------------------- augmenters (compute_forward, TRAIN)
def _sa_table(g, key, B, nf, nt, dev=DEV):
    """Device draw table (layout of include/tsasr_hip.h) from the reference draws stored in the fixture."""
    warp = key + "_c" in g.files
    words = [int(g[key + "_c"][0]) if warp else 0, int(g[key + "_w"][0]) if warp else 0]
    for k in ("_flen", "_fpos", "_tlen", "_tpos"):
        words += g[key + k].reshape(-1).tolist()
    t = torch.tensor(words, dtype=torch.int32, device=dev)
    assert t.numel() == 2 + 2 * B * (nf + nt)
    return t


@pytest.mark.parametrize("name,nf,nt,zero", [("recipe", 2, 2, False), ("zero", 2, 2, True), ("nowarp", 3, 1, False)])
@pytest.mark.parametrize("rep", [0, 1, 2])
def test_spec_augment_kernels_vs_reference_golden(golden, ops, name, nf, nt, zero, rep):
    """tsasr_specaug_apply with the REFERENCE's draws against the reference's output (tests/golden/c1_augment.npz): bicubic time
    warp + frequency masks + time masks with the running global mean as fill. fp32; 1e-5 (cubic taps via FMA, mean via partials)."""
    g = golden["c1_augment"]
    x = T(g["sa_x"]).to(DEV)
    y = ops.spec_augment_apply(x, _sa_table(g, f"sa_{name}_{rep}", 4, nf, nt), nf, nt, zero)
    close(y, g[f"sa_{name}_{rep}_y"], 1e-5, 1e-5)
    assert torch.equal(x.cpu(), T(g["sa_x"]))                 # the input is left untouched


def test_spec_augment_module_draws_and_graph_replay(nn_, ops):
    """nnet.SpecAugment end to end: (1) the device draws respect the reference's ranges (SB/lobes/augment.py:131-136,173-180) and
    their empirical means match the uniform laws; (2) output == oracle.spec_augment fed with those very draws (fp32 and bf16 io);
    (3) too-short inputs skip the warp; (4) captured into a hipGraph, every replay draws new numbers."""
    from oracle import tsasr_ref as R
    aug = nn_.SpecAugment(time_warp=True, time_warp_window=5, freq_mask=True, n_freq_mask=2, time_mask=True, n_time_mask=2,
                          replace_with_zero=False, freq_mask_width=30, time_mask_width=20).to(DEV)
    B, Tn, Fq = 64, 300, 80
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Tn, Fq, generator=g)
    cs, ws, fl, tl, fp_, tp = [], [], [], [], [], []
    for _ in range(40):
        ops.begin_step(torch.device(DEV))
        p = aug.draw(x.to(DEV)).cpu().numpy()
        c, w = int(p[0]), int(p[1])
        flen, fpos, tlen, tpos = (p[2 + i * 2 * B: 2 + (i + 1) * 2 * B].reshape(B, 2) for i in range(4))
        assert 5 <= c < Tn - 5 and c - 5 + 1 <= w <= c + 5
        assert flen.min() >= 0 and flen.max() < 30 and tlen.min() >= 0 and tlen.max() < 20
        assert fpos.min() >= 0 and fpos.max() < max(1, Fq - flen.max()) and tpos.min() >= 0 and tpos.max() < max(1, Tn - tlen.max())
        cs.append(c); ws.append(w - c); fl.append(flen); tl.append(tlen); fp_.append(fpos); tp.append(tpos)
    assert len(set(cs)) > 20                                             # fresh numbers per step
    assert abs(np.mean(fl) - 14.5) < 0.6 and abs(np.mean(tl) - 9.5) < 0.4     # U{0..29}, U{0..19}: sd/sqrt(5120) = 0.12 / 0.08
    assert abs(np.mean(tp) - (Tn - 19 - 1) / 2) < 6 and abs(np.mean(ws) - 0.5) < 1.5
    # (2) same draws through the oracle
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 4e-2)):
        xd = x.to(DEV).to(dtype)
        p = aug.draw(xd)
        y = aug(xd, params=p)
        pc = p.cpu().numpy()
        ref = R.spec_augment(xd.float().cpu(), int(pc[0]), int(pc[1]), *(pc[2 + i * 2 * B: 2 + (i + 1) * 2 * B].reshape(B, 2) for i in range(4)),
                             window=5, replace_with_zero=False)
        assert y.dtype == dtype and y.shape == x.shape
        close(y, ref, tol, tol)
    # (3) time - window <= window: no warp, c = w = 0
    xs = torch.randn(2, 10, 80, generator=g).to(DEV)
    only_warp = nn_.SpecAugment(time_warp=True, time_warp_window=5, freq_mask=False, time_mask=False)
    assert only_warp.draw(xs)[:2].tolist() == [0, 0] and torch.equal(only_warp(xs), xs)
    # (4) graph capture: the stream id is frozen, the device step counter moves
    xd = x.to(DEV)
    ops.begin_step(torch.device(DEV))
    aug(xd)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.begin_step(torch.device(DEV))
        p_static = aug.draw(xd)
        y_static = ops.spec_augment_apply(xd, p_static, 2, 2, False)
    seen = set()
    for _ in range(5):
        graph.replay()
        torch.cuda.synchronize()
        pc = p_static.cpu().numpy()
        seen.add(tuple(pc[:6].tolist()))
        ref = R.spec_augment(x, int(pc[0]), int(pc[1]), *(pc[2 + i * 2 * B: 2 + (i + 1) * 2 * B].reshape(B, 2) for i in range(4)),
                             window=5, replace_with_zero=False)
        close(y_static, ref, 1e-5, 1e-5)
    assert len(seen) == 5


@pytest.mark.parametrize("speed", [95, 105, 50])
def test_resample_kernel_vs_reference_golden(golden, nn_, speed):
    """nnet.Resample (one HIP launch) against the reference's Resample output, incl. the partial windows at both signal ends;
    plus the reference's own unit test (half speed of a sine ~ every second sample, tests/unittests/test_augment.py:110-113)."""
    g = golden["c1_augment"]
    rs = nn_.Resample(16000, 16000 * speed // 100)
    y = rs(T(g["sp_x"]).to(DEV))
    assert y.shape == g[f"sp_{speed}_y"].shape
    close(y, g[f"sp_{speed}_y"], 1e-6, 1e-5)
    y3 = rs(T(g["sp_x"]).t().contiguous()[None].to(DEV))              # [1, L, 3 channels]
    close(y3[0].t(), g[f"sp_{speed}_y"], 1e-6, 1e-5)
    if speed == 50:
        sine = torch.sin(torch.arange(16000.0)).unsqueeze(0).to(DEV)
        half = nn_.SpeedPerturb(16000, speeds=[50]).to(DEV)(sine)
        close(half, g["sp_sine_half"], 2e-6, 1e-5)
        assert half.allclose(sine[:, ::2], atol=3e-1)
    L = 159840 + 77                                                    # BASELINE utterance length, ragged against the 20-sample unit
    from oracle import tsasr_ref as R
    wav = torch.randn(2, L, generator=torch.Generator().manual_seed(1)) * 0.1
    close(rs(wav.to(DEV)), R.resample(wav, 16000, rs.new_freq), 1e-6, 1e-5)


def test_dropout_add_with_outer_dropout(ops):
    """Dropout(r + Dropout(y)) in one pass (the tail of a front-end ConvBlock, SB/lobes/models/convolution.py:260-266): both masks
    are independent counter streams, the forward composes them, the backward regenerates both (fp32 io: exact arithmetic)."""
    g = torch.Generator().manual_seed(11)
    x, r = torch.randn(40, 300, 128, generator=g), torch.randn(40, 300, 128, generator=g)
    xg, rg = x.to(DEV).requires_grad_(), r.to(DEV).requires_grad_()
    close(ops.dropout_add(xg, None, rg, 1.0, 0.1, False, outer_p=0.1), x + r, 1e-6)         # eval: identity
    close(ops.dropout_add(xg, None, rg, 1.0, 0.0, True, outer_p=0.0), x + r, 1e-6)          # p = 0
    out = ops.dropout_add(xg, None, rg, 1.0, 0.1, True, outer_p=0.1)
    keep2 = out != 0
    n = out.numel()
    assert abs(keep2.float().mean().item() - 0.9) < 5 * (0.09 / n) ** 0.5 + 1e-5
    inner = torch.where(keep2, out * (58982 / 65536) - rg.detach(), torch.zeros_like(out))   # undo the outer scale (p quantised to 1/65536)
    big = xg.detach().abs() > 1e-2                             # |x| large enough to tell "kept" from "dropped" through fp32 rounding of r
    keep1 = inner.abs() > 0.5 * xg.detach().abs()
    sel = keep2 & keep1 & big
    close(inner[sel], (xg.detach() / (58982 / 65536))[sel], 2e-5, 1e-4)
    frac1 = (keep1 & keep2 & big).float().sum().item() / (keep2 & big).float().sum().item()
    assert abs(frac1 - 0.9) < 2e-3                                                           # inner mask independent of the outer one
    out.backward(torch.ones_like(out))
    s2 = 65536 / 58982
    close(rg.grad, keep2.float() * s2, 1e-6)
    close(xg.grad[keep2 & big], (keep1.float() * s2 * s2)[keep2 & big], 1e-5)
    assert torch.all(xg.grad[~keep2] == 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(160000, 128), (777, 256), (5, 2048)])
def test_colsum_kernel(ops, dtype, M, N):
    """tsasr_colsum (bias gradients of the GEMM-shaped front-end convolutions) against a float64 column sum."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g).to(dtype)
    ref = x.double().sum(0)
    out = ops.colsum(x.to(DEV))
    tol = 2e-3 * (M ** 0.5)
    close(out, ref.float(), tol, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layer_norm_res_sums_both_gradients(ops, dtype):
    """(LayerNorm(x), x) with the residual pass-through: forward values and the summed gradient equal LayerNorm + an explicit add."""
    g = torch.Generator().manual_seed(3)
    M, D = 300, 256
    x = torch.randn(M, D, generator=g).to(dtype)
    w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    dy, dr = torch.randn(M, D, generator=g).to(dtype), torch.randn(M, D, generator=g).to(dtype)
    res = {}
    for which in ("res", "plain"):
        xg = x.to(DEV).requires_grad_()
        wg, bg = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        if which == "res":
            y, xa = ops.layer_norm_res(xg, wg, bg, 1e-5)
        else:
            y, xa = ops.layer_norm(xg, wg, bg, 1e-5), xg
        torch.autograd.backward([y, xa], [dy.to(DEV), dr.to(DEV)])
        res[which] = (y.detach(), xg.grad, wg.grad, bg.grad)
    lo = dtype == torch.bfloat16
    assert torch.equal(res["res"][0], res["plain"][0])
    close(res["res"][1], res["plain"][1], 3e-2 if lo else 1e-6, 1e-2 if lo else 1e-6)      # one rounding less on the fused path
    close(res["res"][2], res["plain"][2], 1e-4, 1e-5)
    close(res["res"][3], res["plain"][3], 1e-4, 1e-5)
    # only the residual path has a gradient
    xg = x.to(DEV).requires_grad_()
    y, xa = ops.layer_norm_res(xg, w.to(DEV), b.to(DEV), 1e-5)
    xa.backward(dr.to(DEV))
    assert torch.equal(xg.grad, dr.to(DEV))


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 256), (torch.bfloat16, 144), (torch.float32, 144), (torch.float32, 256), (torch.bfloat16, 512)])
@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("with_dy", [True, False])
def test_add_layer_norm2_equals_the_two_launches_it_replaces(ops, dtype, D, p, with_dy):
    """The layer seam norm2 -> next layer's first LayerNorm (Conformer.py:194-217,259; after the last layer the encoder's final norm,
    eps 1e-6: models/conformer.py:223-233) in ONE launch each way (tsasr_add_layernorm2_fwd/bwd) against the two launches it replaces
    (tsasr_add_layernorm_* then tsasr_layernorm_*, themselves pinned to the oracle by the block tests above): y, z and the data
    gradients to the last bits of the row sums (same dropout seed, ragged valid lengths; bf16: < 0.2 % of the elements differ, by one
    ulp), parameter gradients to summation order. with_dy=False: y has no
    other reader (the last layer)."""
    B, Tn = 3, 50
    g = torch.Generator().manual_seed(D + int(p * 10))
    mk = lambda *shape: torch.randn(*shape, generator=g)  # noqa: E731
    x0, r0 = mk(B, Tn, D).to(dtype), mk(B, Tn, D).to(dtype)
    par0 = [mk(D) * 0.1, 1 + 0.1 * mk(D), 0.1 * mk(D), 1 + 0.1 * mk(D), 0.1 * mk(D)]   # bias, g1, b1, g2, b2
    wz, wy = mk(B, Tn, D).to(DEV), mk(B, Tn, D).to(DEV)
    valid = torch.tensor([Tn, 31, 7], dtype=torch.int32, device=DEV)
    out = []
    for fused in (False, True):
        x, r = x0.clone().to(DEV).requires_grad_(), r0.clone().to(DEV).requires_grad_()
        par = [t.clone().to(DEV).requires_grad_() for t in par0]
        if fused:
            y, z = ops._AddLayerNorm2Fn.apply(x, par[0], r, par[1], par[2], par[3], par[4], 0.5, p, 4321, valid, Tn, 1e-5, 1e-6)
        else:
            _, y0 = ops._AddLayerNormFn.apply(x, par[0], r, par[1], par[2], 0.5, p, 4321, valid, Tn, 1e-5)
            z, y = ops._LayerNormResFn.apply(y0, par[3], par[4], 1e-6)
        loss = (z.float() * wz).sum() + ((y.float() * wy).sum() if with_dy else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        out.append((y.detach(), z.detach(), x.grad, r.grad, [t.grad for t in par]))
    (y_a, z_a, dx_a, dr_a, pg_a), (y_b, z_b, dx_b, dr_b, pg_b) = out
    assert torch.equal(y_a, y_b) and torch.equal(z_a, z_b)
    # backward: the row sums of the kernels associate differently (last fp32 bits; in bf16 that flips the odd rounding: one ulp)
    rt, at = (1e-5, 1e-6) if dtype == torch.float32 else (2 ** -7, 2 ** -9)
    torch.testing.assert_close(dx_b.float(), dx_a.float(), rtol=rt, atol=at)
    torch.testing.assert_close(dr_b.float(), dr_a.float(), rtol=rt, atol=at)
    if dtype == torch.bfloat16:
        frac = max(float((dx_b != dx_a).float().mean()), float((dr_b != dr_a).float().mean()))
        assert frac < 2e-3, frac
    for a, b_, name in zip(pg_a, pg_b, ("bias", "gamma", "beta", "gamma2", "beta2")):
        np.testing.assert_allclose(b_.float().cpu().numpy(), a.float().cpu().numpy(), rtol=2e-4, atol=2e-4 * float(a.abs().max()), err_msg=name)
