"""Fused ConvBlock kernels (csrc/frontend_block.hip) against the oracle's ConvBlock (oracle/tsasr_ref.py conv_block, which follows
SB/lobes/models/convolution.py:187-266) - forward and every gradient, with and without dropout.

With dropout the masks are a function of (seed, element index) that only the kernels know; the test recovers them from two extra
forward calls with doctored LayerNorm parameters (1x1 branch = a constant: the output is non-zero exactly where the outer mask keeps;
1x1 branch = 0: non-zero where both masks keep) and hands them to the oracle expression, so forward and backward are checked against
the same masked function the kernels claim to compute.
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu
DEV = "cuda"
SLOPE, EPS = 0.01, 1e-5


def _mods():
    return importlib.import_module("ts-asr_amd.ops"), importlib.import_module("oracle.tsasr_ref")


def _params(cin, co, fo, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)  # noqa: E731
    return dict(
        w1=r(co, cin, 3, 3, scale=1.0 / (3.0 * cin ** 0.5)), b1=r(co, scale=0.1), w2=r(co, cin, 1, 1, scale=1.0 / cin ** 0.5), b2=r(co, scale=0.1),
        g1=1.0 + r(fo, co, scale=0.1), be1=r(fo, co, scale=0.1), g2=1.0 + r(fo, co, scale=0.1), be2=r(fo, co, scale=0.1),
    )


def _oracle_block(R, x, P, padding, keep_in=None, keep_out=None, p=0.0):
    """x [B,T,F,Cin] fp32 -> ConvBlock output; masks (bool, shape of the output) applied where given."""
    xt = x.transpose(1, -1)
    y = R._conv2d_sb(xt, P["w1"], P["b1"], 2, padding).transpose(1, -1)
    y = F.leaky_relu(F.layer_norm(y, P["g1"].shape, P["g1"], P["be1"], EPS), SLOPE)
    r = R._conv2d_sb(xt, P["w2"], P["b2"], 2, "same").transpose(1, -1)
    r = F.layer_norm(r, P["g2"].shape, P["g2"], P["be2"], EPS)
    if keep_in is not None:
        y = y * keep_in / (1.0 - p)
    s = y + r
    if keep_out is not None:
        s = s * keep_out / (1.0 - p)
    return s


def _hip_block(ops, x, P, padding, p, seeds, training=True):
    """Same call ConvBlock.forward makes, with explicit seeds."""
    causal = padding == "causal"
    conv_params, ln_params = (P["w1"], P["b1"], P["w2"], P["b2"]), (P["g1"], P["be1"], P["g2"], P["be2"])
    pp = p if training else 0.0
    if x.shape[-1] == 1:
        return ops._FrontendBlockFn.apply(x.squeeze(-1), None, None, conv_params, ln_params, 0, 0, causal, SLOPE, EPS, pp, seeds[0], pp, seeds[1],
                                          *conv_params, *ln_params)
    y1, y2 = ops._FrontendConvFn.apply(x, P["w1"], P["b1"], P["w2"], P["b2"], causal)
    return ops._FrontendBlockFn.apply(None, y1, y2, None, ln_params, x.shape[1], x.shape[2], causal, SLOPE, EPS, pp, seeds[0], pp, seeds[1],
                                      None, None, None, None, *ln_params)


def _quantile(p, thr_bits=16):
    return min(65535, int(p * 65536 + 0.5)) / 65536.0


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))  # noqa


CASES = [  # (cin, B, T, F, padding)
    (1, 3, 37, 80, "same"), (1, 2, 64, 80, "causal"), (1, 2, 21, 38, "same"), (1, 2, 16, 24, "causal"),
    (128, 2, 19, 40, "same"), (128, 2, 24, 40, "causal"),
]


@pytest.mark.parametrize("cin,B,T,Fq,padding", CASES)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fused_convblock_vs_oracle(cin, B, T, Fq, padding, dtype, p):
    ops, R = _mods()
    co, fo, to = 128, (Fq - 1) // 2 + 1, (T - 1) // 2 + 1
    assert ops.C.lib().tsasr_frontend_block_supported(fo, co)
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    P = _params(cin, co, fo, seed=cin + T)
    x = torch.randn(B, T, Fq, cin, generator=torch.Generator().manual_seed(5))
    if dtype == "bf16":
        x = x.bfloat16().float()          # both sides see the same (bf16-representable) input
    seeds = (0x1234567, 0x7654321)
    Pd = {k: v.to(DEV).requires_grad_(True) for k, v in P.items()}
    xd = x.to(DEV).to(tdt).requires_grad_(cin > 1)

    keep_in = keep_out = None
    if p > 0:
        with torch.no_grad():
            Pc = dict(Pd, g2=torch.zeros_like(Pd["g2"]), be2=torch.full_like(Pd["be2"], 64.0))
            keep_out = _hip_block(ops, xd, Pc, padding, p, seeds).float().cpu() != 0
            Pz = dict(Pd, g2=torch.zeros_like(Pd["g2"]), be2=torch.zeros_like(Pd["be2"]),
                      g1=torch.zeros_like(Pd["g1"]), be1=torch.ones_like(Pd["be1"]))          # 3x3 branch = constant 1 before its dropout
            both = _hip_block(ops, xd, Pz, padding, p, seeds).float().cpu() != 0
        keep_in = both | ~keep_out           # where the outer mask drops, the inner bit does not matter
        q = _quantile(p)
        n = keep_out.numel()
        assert abs(float(keep_out.float().mean()) - (1 - q)) < 4 * (q * (1 - q) / n) ** 0.5 + 1e-3
        assert abs(float(both.float().mean()) - (1 - q) ** 2) < 4 * (0.25 / n) ** 0.5 + 1e-3
        p_eff = q
    else:
        p_eff = 0.0

    out = _hip_block(ops, xd, Pd, padding, p, seeds)
    assert out.shape == (B, to, fo, co) and out.dtype == tdt
    dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    if dtype == "bf16":
        dout = dout.bfloat16().float()
    out.backward(dout.to(DEV).to(tdt))
    ops.reduce_flush()
    torch.cuda.synchronize()

    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(cin > 1)
    ref = _oracle_block(R, xr, Pr, padding, keep_in, keep_out, p_eff)
    ref.backward(dout)

    # budgets: fp32 = summation-order noise; bf16 = one rounding of the output (and, for the wide block, bf16 conv outputs / GEMMs)
    fwd_tol = 2e-5 if dtype == "fp32" else (4e-3 if cin == 1 else 8e-3)
    # (wide block in bf16: y1 is a bf16-rounded GEMM output, so LeakyReLU' flips between 1 and 0.01 for pre-activations within a
    # rounding step of 0 - measured 3.7-4.0e-2 on w1/b1/be1 at these tiny sizes, 2e-3 on the 1x1 branch that has no activation)
    grad_tol = 2e-4 if dtype == "fp32" else (1.5e-2 if cin == 1 else 6e-2)
    assert rel_l2(out, ref.detach()) < fwd_tol
    worst = {}
    for k in P:
        worst[k] = rel_l2(Pd[k].grad, Pr[k].grad)
    if cin > 1:
        worst["x"] = rel_l2(xd.grad, xr.grad)
    bad = {k: v for k, v in worst.items() if not v < grad_tol}
    assert not bad, f"gradient rel-L2 over budget {grad_tol}: {bad} (all: {worst})"


def test_fused_convblock_eval_and_determinism():
    """training=False ignores p; two identical calls give identical bits (fixed-order reductions)."""
    ops, R = _mods()
    P = {k: v.to(DEV) for k, v in _params(1, 128, 40, seed=3).items()}
    x = torch.randn(2, 50, 80, 1, generator=torch.Generator().manual_seed(1)).to(DEV).bfloat16()
    a = _hip_block(ops, x, P, "same", 0.3, (1, 2), training=False)
    b = _hip_block(ops, x, P, "same", 0.0, (3, 4), training=True)
    assert torch.equal(a, b)
    grads = []
    for _ in range(2):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        o = _hip_block(ops, x, Pg, "same", 0.1, (11, 12))
        o.backward(torch.ones_like(o))
        ops.reduce_flush()
        torch.cuda.synchronize()
        grads.append({k: v.grad.clone() for k, v in Pg.items()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_convblock_module_uses_fused_path():
    """ConvBlock.forward routes through the fused kernels for the YAML's shapes and matches the unfused HIP chain (p = 0)."""
    nnet = importlib.import_module("ts-asr_amd.nnet")
    ops, _ = _mods()
    torch.manual_seed(0)
    blk = nnet.ConvBlock(1, 128, 80).to(DEV)
    x = torch.randn(2, 40, 80, 1, device=DEV).bfloat16()
    prev = nnet._FUSED_CONVBLOCK
    try:
        nnet._FUSED_CONVBLOCK = True
        a = blk(x, 0.0, True)
        nnet._FUSED_CONVBLOCK = False
        b = blk(x, 0.0, True)
    finally:
        nnet._FUSED_CONVBLOCK = prev
    assert a.shape == b.shape
    assert rel_l2(a, b) < 6e-3     # the unfused chain rounds y, r and both LayerNorm outputs to bf16 on the way


@pytest.mark.parametrize("B,T,F,Ci,causal", [(2, 37, 21, 128, False), (2, 37, 21, 128, True), (1, 500, 40, 128, False), (3, 10, 8, 64, True), (1, 2, 2, 128, False),
                                             (2, 3, 3, 64, False), (4, 124, 40, 64, False), (2, 45, 40, 64, True), (16, 500, 40, 128, False)])
def test_conv_implicit_gemm_equals_im2col_path(B, T, F, Ci, causal):
    """Front-end block 2's convolutions as implicit GEMMs (csrc/gemm.hip conv_s2_fwd / conv_s2_wgrad: the ring kernels' loader waves gather
    the 3x3 patch rows, padding rule folded into the address - SB/nnet/CNN.py:629-711 reflect / causal) against the im2col + GEMM path
    they replace, which the golden-vector tests above pin to the reference: same bf16 operands, fp32 accumulation in another order.
    Outputs, data gradient (conv_s2_dgrad: one gathered GEMM per class of input pixels, against dy . Wm + col2im), both filter gradients and
    both bias gradients; odd and even sizes, reflection at both ends (an index reached by four taps at T = F = 3), causal zero padding, a
    single output position, Ci = 64."""
    ops = importlib.import_module("ts-asr_amd.ops")
    Co = 128
    g = torch.Generator().manual_seed(T * 3 + F + Ci)
    x = torch.randn(B, T, F, Ci, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(DEV)
    w2 = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).to(DEV)
    b1, b2 = torch.randn(Co, generator=g).to(DEV), torch.randn(Co, generator=g).to(DEV)
    To, Fo = (T - 1) // 2 + 1, (F - 1) // 2 + 1
    d1 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16).to(DEV)
    d2 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    for implicit in (True, False):
        ops.CONV_IMPLICIT = implicit
        try:
            leaves = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
            y1, y2 = ops._FrontendConvFn.apply(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], causal)
            torch.autograd.backward([y1, y2], [d1, d2])
            res[implicit] = [y1.detach().float(), y2.detach().float()] + [t.grad.float() for t in leaves]
        finally:
            ops.CONV_IMPLICIT = True
    for a, b_, name in zip(res[True], res[False], ("y1", "y2", "dx", "dw1", "db1", "dw2", "db2")):
        rel = float((a - b_).norm() / (b_.norm() + 1e-30))
        assert rel < (4e-3 if name in ("y1", "y2", "dx") else 2e-4), (name, rel)      # bf16 outputs round once each way; fp32 filter gradients


@pytest.mark.parametrize("B,T,F,Ci,causal", [(2, 9, 6, 64, False), (1, 7, 5, 128, True), (3, 4, 4, 64, False)])
def test_conv_data_gradient_vs_definition(B, T, F, Ci, causal):
    """conv_s2_dgrad against the transpose of the convolution written out from its definition in float64 (SB/nnet/CNN.py:629-711: reflect /
    causal padding as csrc/frontend.hip src_index; the 1x1 stride-2 branch reads x[2t', 2f']): same bf16 operands, one rounding of the result."""
    ops = importlib.import_module("ts-asr_amd.ops")
    Co = 128
    g = torch.Generator().manual_seed(B * 100 + T * 7 + F)
    x = torch.randn(B, T, F, Ci, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5))
    w2 = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5)
    b1, b2 = torch.randn(Co, generator=g), torch.randn(Co, generator=g)
    To, Fo = (T - 1) // 2 + 1, (F - 1) // 2 + 1
    d1 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16)
    d2 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16)
    xl = x.clone().requires_grad_()
    y1, y2 = ops._FrontendConvFn.apply(xl, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), causal)
    torch.autograd.backward([y1, y2], [d1.to(DEV), d2.to(DEV)])

    def src(o, k, n, mode):
        if mode == 1:
            i = 2 * o + k - 2
            return -1 if i < 0 else i
        i = 2 * o + k - 1
        if mode == 0:
            i = -i if i < 0 else i
            return 2 * (n - 1) - i if i >= n else i
        return -1 if (i < 0 or i >= n) else i

    tm, fm = (1, 2) if causal else (0, 0)
    w1b = w1.to(torch.bfloat16).double().numpy()      # [Co, Ci, kF, kT]
    w2b = w2.to(torch.bfloat16).double().numpy().reshape(Co, Ci)
    d1n, d2n = d1.double().numpy(), d2.double().numpy()
    ref = np.zeros((B, T, F, Ci))
    for to in range(To):
        for fo in range(Fo):
            for kt in range(3):
                for kf in range(3):
                    ti, fi = src(to, kt, T, tm), src(fo, kf, F, fm)
                    if ti >= 0 and fi >= 0:
                        ref[:, ti, fi, :] += d1n[:, to, fo, :] @ w1b[:, :, kf, kt]
            ref[:, 2 * to, 2 * fo, :] += d2n[:, to, fo, :] @ w2b
    got = xl.grad.double().cpu().numpy()
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max() and np.linalg.norm(got - ref) / np.linalg.norm(ref) < 3e-3      # one bf16 rounding
