"""GPU parity at BASELINE.json configs[4] sizes: causal long-form, T = 16000 mel frames -> T' = 4000 encoder frames, B = 1 per GPU,
U = 1920 tokens (lattice 4000 x 1921: 31 columns per lane), and the build extension chunk = 40 (block-causal attention; the
reference has no chunked attention: models/conformer.py:279-280 only builds the look-ahead mask).
Every product call goes through the C-ABI; the checker is the CPU oracle at full size (fp32 / float64, seconds per case)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from oracle import rnnt_ref as RR  # noqa: E402
from oracle import tsasr_ref as R  # noqa: E402
from oracle.golden_recipe import det_tensor, load_det_weights  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TP, U1 = 4000, 1921


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def rel_l2(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).norm() / (b.detach().float().norm() + 1e-30))


@pytest.fixture(scope="module")
def nn_():
    return importlib.import_module("ts-asr_amd.nnet")


# ---------------------------------------------------------------------------------------------- attention at T' = 4000
@pytest.mark.parametrize("chunk", [1, 40])
def test_relpos_mha_T4000_causal_vs_oracle(nn_, chunk):
    """RelPosMHAXL (in/pos/out projections on the HIP GEMM, fused causal attention forward + the four backward kernels) at
    B=1, T'=4000, D=256, H=4 in bf16 against oracle.relpos_mha (fp32, CPU): output and every gradient. chunk=1: the reference's
    look-ahead mask (SB/nnet/attention.py:607-616 with get_lookahead_mask); chunk=40: the build's block-causal extension."""
    D, H = 256, 4
    nn_.set_compute_dtype(torch.bfloat16)
    mha = load_det_weights(nn_.RelPosMHAXL(embed_dim=D, num_heads=H, dropout=0.0, mask_pos_future=True), "blk.mha.").to(DEV).eval()
    x = T(det_tensor("long.x", (1, TP, D), 1.0))
    probe = T(det_tensor("long.probe", (1, TP, D), 1.0))
    pe = nn_.RelPosEncXL(D).to(DEV)(x.to(DEV))
    xg = x.to(DEV).to(torch.bfloat16).requires_grad_()
    out = mha(xg, xg, xg, pe, return_attn_weights=False, causal=chunk)
    (out.float() * probe.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    # oracle, fp32 on the host
    sd = {"m." + k: v.detach().float().cpu().clone().requires_grad_() for k, v in mha.state_dict().items()}
    xo = x.clone().requires_grad_()
    ref = R.relpos_mha(xo, R.relpos_table(TP, D), sd, "m.", H, None, chunk)
    (ref * probe).sum().backward()
    e_out = rel_l2(out, ref)
    errs = {"out": e_out, "dx": rel_l2(xg.grad, xo.grad)}
    for k, p in mha.named_parameters():
        errs[k] = rel_l2(p.grad, sd["m." + k].grad)
    print(f"T'=4000 causal chunk={chunk}:", {k: round(v, 4) for k, v in errs.items()})
    assert e_out < 1.5e-2 and errs["dx"] < 3e-2                     # bf16 activations, bf16 MFMA operands, fp32 softmax / accumulation
    for k, v in errs.items():
        assert v < (1.5e-1 if ("pos_bias" in k or "linear_pos" in k) else 4e-2), (k, v)
    # causality as a property: output frame i must not change when frames beyond its limit change
    lim = 1999 if chunk == 1 else (1999 // 40 + 1) * 40 - 1
    x2 = x.clone()
    x2[:, lim + 1:] += 1.0
    x2g = x2.to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        o2 = mha(x2g, x2g, x2g, pe, return_attn_weights=False, causal=chunk)
    assert torch.equal(o2[:, :2000], out.detach()[:, :2000])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_module_T4000_causal_vs_oracle(nn_, dtype):
    """ConvolutionModule (causal depthwise K=31, Conformer.py:68-71,108-110) at [1, 4000, 256] vs the oracle, forward and gradients."""
    D = 256
    nn_.set_compute_dtype(dtype)
    conv = load_det_weights(nn_.ConvolutionModule(D, 31, True, torch.nn.LeakyReLU, 0.0, causal=True), "blk.conv.").to(DEV).eval()
    x = T(det_tensor("long.x", (1, TP, D), 1.0))
    probe = T(det_tensor("long.probe", (1, TP, D), 1.0))
    xg = x.to(DEV).to(dtype).requires_grad_()
    out = conv(xg)
    (out.float() * probe.to(DEV)).sum().backward()
    sd = {"c." + k: v.detach().float().cpu().clone().requires_grad_() for k, v in conv.state_dict().items()}
    xo = x.clone().requires_grad_()
    ref = R.conv_module(xo, sd, "c.", None, True)
    (ref * probe).sum().backward()
    errs = {"out": rel_l2(out, ref), "dx": rel_l2(xg.grad, xo.grad)}
    for k, p in conv.named_parameters():
        errs[k] = rel_l2(p.grad, sd["c." + k].grad)
    print(f"conv module T'=4000 {dtype}:", {k: round(v, 5) for k, v in errs.items()})
    bf = dtype == torch.bfloat16
    # bf16: two LayerNorm backwards on bf16-rounded activations dominate dx (the probe-weighted gradient passes through 1/sigma twice)
    assert errs["out"] < (1.2e-2 if bf else 2e-5) and errs["dx"] < (8e-2 if bf else 1e-4)
    for k, v in errs.items():
        assert v < (8e-2 if bf else 2e-4), (k, v)
    nn_.set_compute_dtype(torch.bfloat16)
    x2 = x.clone()
    x2[:, 2000:] += 1.0
    with torch.no_grad():
        assert torch.equal(conv(x2.to(DEV).to(dtype))[:, :2000], out.detach()[:, :2000])


# ---------------------------------------------------------------------------------------------- RNN-T lattice, 31 columns per lane
def test_rnnt_lattice_u1921_vs_oracle():
    """U+1 = 1921 lattice columns (K = 31 per lane; configs[1] uses K = 2) on short ragged T against the float64 C oracle: costs, gradients."""
    rn = importlib.import_module("ts-asr_amd.rnnt")
    B, Tn, V = 2, 48, 29
    lg = T(det_tensor("long.logits", (B, Tn, U1, V), 2.0))
    tg = T(np.random.default_rng(5).integers(1, V, size=(B, U1 - 1)).astype(np.int32))
    tl, ul = torch.tensor([48, 31], dtype=torch.int32), torch.tensor([1920, 1500], dtype=torch.int32)
    lgd = lg.to(DEV).requires_grad_()
    costs = rn.rnnt_costs(lgd, tg.to(DEV), tl.to(DEV), ul.to(DEV), 0)
    costs.sum().backward()
    cref, gref = RR.rnnt_costs_grads(lg.numpy(), tg.numpy(), tl.numpy(), ul.numpy(), 0)
    np.testing.assert_allclose(costs.detach().cpu().numpy(), cref, rtol=2e-5)
    g = lgd.grad.cpu().numpy()
    np.testing.assert_allclose(g, gref, atol=3e-5, rtol=max(1e-3, 5e-5 * (Tn + U1)))
    assert np.all(g[1, 31:] == 0) and np.all(g[1, :, 1501:] == 0)
    # this shape splits each lattice over single-wave workgroups that wait for one another (csrc/rnnt.hip MC): the launch has a time-out
    # word in its workspace, raised when a wait runs out (the costs are then NaN); Brain.flush_nonfinite reads it. Shapes whose lattice is
    # one workgroup have none.
    C = importlib.import_module("ts-asr_amd._capi")
    assert C.lib().tsasr_rnnt_loss_error_word_offset(1, 4000, U1) >= 0 and C.lib().tsasr_rnnt_loss_error_word_offset(32, 250, 121) == -1
    one = rn.rnnt_costs(lgd[:1].detach(), tg[:1].to(DEV), tl[:1].to(DEV), ul[:1].to(DEV), 0)      # B = 1: the split form
    assert torch.isfinite(one).all() and abs(float(one[0]) - float(cref[0])) < 2e-5 * abs(float(cref[0]))
    assert (len(rn._LATTICE_ERR) > 0) == (C.lib().tsasr_rnnt_loss_error_word_offset(1, Tn, U1) >= 0) and rn.lattice_timeouts() == 0


def test_longform_joint_and_loss_full_size_properties():
    """configs[4] lattice B=1, T'=4000, U+1=1921, J=640, V=29 (983 MB of logits): the loss on the HIP logits equals the float64 C oracle,
    dlogits match it and sum to zero over v, enc/dec gradients are finite, two runs are bitwise equal."""
    rn = importlib.import_module("ts-asr_amd.rnnt")
    J, V = 640, 29
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(1, TP, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
    dec = torch.randn(1, U1, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
    W = (torch.randn(V, J, generator=g) / J ** 0.5).to(DEV).requires_grad_()
    b = torch.zeros(V, device=DEV).requires_grad_()
    tg = torch.randint(1, V, (1, U1 - 1), generator=g, dtype=torch.int32)
    tl, ul = torch.tensor([TP], dtype=torch.int32), torch.tensor([U1 - 1], dtype=torch.int32)
    outs = []
    for _ in range(2):
        for p in (enc, dec, W, b):
            p.grad = None
        logits = rn.fused_joint_logits(enc, dec, W, b, 0.01, tl.to(DEV), ul.to(DEV))
        logits.retain_grad()
        costs = rn.rnnt_costs(logits, tg.to(DEV), tl.to(DEV), ul.to(DEV), 0)
        costs.mean().backward()
        outs.append((costs.detach().clone(), enc.grad.clone(), dec.grad.clone(), W.grad.clone()))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)
    cref, gref = RR.rnnt_costs_grads(np.ascontiguousarray(logits.detach().cpu().numpy()), tg.numpy(), tl.numpy(), ul.numpy(), 0)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), cref, rtol=3e-5)
    dl = logits.grad.cpu().numpy()
    np.testing.assert_allclose(dl, gref, atol=3e-5, rtol=max(1e-3, 5e-5 * (TP + U1)))
    print('max |sum_v dlogits|', float(np.abs(dl.sum(-1)).max()))
    np.testing.assert_allclose(dl.sum(-1), 0, atol=3e-3)   # fp32 alpha/beta over a perimeter of 5921 cells (measured 1.0e-3; the oracle is float64)
    assert all(bool(torch.isfinite(t).all()) for t in outs[0])


# ---------------------------------------------------------------------------------------------- whole model, long-form
@pytest.mark.parametrize("chunk", [0, 40])
def test_longform_model_loss_vs_oracle(chunk):
    """The whole path at T = 16000 mel frames (-> 4000), U = 1920, B = 1, causal encoder + causal front-end padding (the reference's
    --causal_encoder True --frontend_padding causal), configs[0] model width (2 layers, d_model 144) so that the CPU oracle finishes in
    seconds: loss of the bf16 HIP path vs the oracle's, and one optimizer step runs. chunk=40: the block-causal extension."""
    from oracle.golden_recipe import CFG1
    bm = importlib.import_module("ts-asr_amd.batch")
    brain, h = entry._config1_brain(DEV, "bf16", causal_encoder=True, frontend_padding="causal", attention_chunk_size=chunk, input_is_feats=True)
    brain.modules.train()
    batch = bm.synthetic_batch(1, 16000, 400, 1920, feats=True, seed=3)
    sd = {f"{n}.{k}": v.detach().cpu().float().clone() for n, m in brain.modules.items() for k, v in m.state_dict().items()}
    loss = brain.fit_batch(batch)
    cb = {"mixed_feats": batch.mixed_sig.data.cpu(), "mixed_lens": batch.mixed_sig.lengths.cpu(), "enroll_feats": batch.enroll_sig.data.cpu(),
          "enroll_lens": batch.enroll_sig.lengths.cpu(), "tokens_bos": batch.tokens_bos.data.cpu(), "tokens_bos_lens": batch.tokens_bos.lengths.cpu()}
    with torch.no_grad():
        logits = R.compute_forward(cb, sd, CFG1, "cat", causal=(max(chunk, 1)), frontend_padding="causal", from_feats=True)
    ref, _ = RR.transducer_loss_ref(logits.numpy(), batch.tokens.data.cpu().numpy(), batch.mixed_sig.lengths.cpu().numpy(),
                                    batch.tokens.lengths.cpu().numpy(), 0, "mean")
    print(f"long-form loss chunk={chunk}: hip {float(loss):.4f} oracle {ref:.4f}")
    assert float(loss) == pytest.approx(ref, rel=3e-2)
    assert brain.optimizer_step == 1 and brain.flush_nonfinite() == 0


# ---------------------------------------------------------------------------------------------- key-split forward == unsplit forward
@pytest.mark.parametrize("B,Tn,H,Dh,causal,pdrop", [(1, 4000, 4, 64, 1, 0.0), (2, 1500, 4, 64, 0, 0.1), (1, 2077, 2, 36, 40, 0.1),
                                                    (3, 1100, 2, 64, 1, 0.0)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_attention_forward_key_split_equals_unsplit(B, Tn, H, Dh, causal, pdrop, dtype):
    """tsasr_relpos_attn_fwd_ws with a workspace splits the keys of long sequences over workgroups and merges the partial softmaxes;
    with workspace = NULL the same entry runs one workgroup per query block. Same inputs, same dropout seed (the mask is a function of
    (seed, b, h, i, j)): outputs agree to fp32 re-association, log-sum-exp rows too; ragged key lengths, padded head dim, block-causal."""
    C = importlib.import_module("ts-asr_amd._capi")
    D = H * Dh
    g = torch.Generator().manual_seed(Tn + Dh)
    qkv = torch.randn(B, Tn, 3 * D, generator=g).to(dtype).to(DEV)
    pk = torch.randn(2 * Tn - 1, D, generator=g).to(dtype).to(DEV)
    u, v = (torch.randn(H * Dh, generator=g) * 0.3).to(DEV), (torch.randn(H * Dh, generator=g) * 0.3).to(DEV)
    lens = torch.tensor([Tn, max(1, Tn // 3), max(1, Tn - 7)][:B], dtype=torch.int32, device=DEV)
    nbytes = C.lib().tsasr_relpos_attn_fwd_workspace_bytes(B, Tn, H)
    assert nbytes > 0, "these shapes are meant to take the split path"
    res = []
    for ws in (None, torch.empty(nbytes, dtype=torch.uint8, device=DEV)):
        out = torch.full((B, Tn, D), float("nan"), dtype=dtype, device=DEV)
        lse = torch.full((B, H, Tn), float("nan"), dtype=torch.float32, device=DEV)
        C.check(C.lib().tsasr_relpos_attn_fwd_ws(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, Tn, H, Dh,
                                                 1.0 / D ** 0.5, causal, pdrop, 1234, None, C.io_dtype(qkv), C.ptr(ws),
                                                 nbytes if ws is not None else 0, C.stream_ptr()), "attn")
        torch.cuda.synchronize()
        res.append((out.float().cpu(), lse.cpu()))
    (o0, l0), (o1, l1) = res
    assert torch.isfinite(o1).all() and torch.isfinite(l1).all()
    # probabilities enter the P.V MFMA as bf16(exp(s - running max)): the running max differs between the two walks, so the roundings do.
    # bf16 with Dh = 64 (round 5): the split path is the everything-in-LDS kernel per chunk of 256 keys (csrc/attention_short.hip, CHUNK), whose
    # positional products cross LDS as fp16 (2^-11 relative on |BD| ~ 10 raw = ~3e-4 on a scaled score): measured 3.1e-3 / 4.1e-4 at worst
    chunked = dtype == torch.bfloat16 and Dh == 64
    assert float((o1 - o0).norm() / o0.norm()) < (5e-3 if chunked else 3e-3)
    assert float((l1 - l0).abs().max()) < (1e-3 if chunked else 1e-4)


@pytest.mark.parametrize("B,Tn,H,causal,pdrop", [(1, 1500, 4, 1, 0.1), (2, 777, 2, 40, 0.0), (1, 4000, 4, 1, 0.0)])
def test_attention_backward_ignores_workspace_garbage(B, Tn, H, causal, pdrop):
    """Under a look-ahead mask the streaming backward clears only a BAND of its two [B,H,T,T] scratch matrices (what the key-major pass can read
    beyond the query-major pass's writes, csrc/attention.hip attn_zero_band_kernel) instead of all of them. The same call on a workspace full
    of NaN bits and on a workspace of zeros: every gradient finite and bit-identical."""
    C = importlib.import_module("ts-asr_amd._capi")
    Dh = 64
    D = H * Dh
    g = torch.Generator().manual_seed(Tn)
    qkv = (torch.randn(B, Tn, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    pk = (torch.randn(2 * Tn - 1, D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    u, v = (torch.randn(D, generator=g) * 0.1).to(DEV), (torch.randn(D, generator=g) * 0.1).to(DEV)
    lens = torch.tensor([Tn, max(1, Tn - 13)][:B], dtype=torch.int32, device=DEV)
    dout = torch.randn(B, Tn, D, generator=g).to(torch.bfloat16).to(DEV)
    out, lse = torch.empty(B, Tn, D, dtype=torch.bfloat16, device=DEV), torch.empty(B, H, Tn, device=DEV)
    scale = 1.0 / D ** 0.5
    nf = C.lib().tsasr_relpos_attn_fwd_workspace_bytes(B, Tn, H)
    wsf = torch.empty(max(nf, 16), dtype=torch.uint8, device=DEV)
    C.check(C.lib().tsasr_relpos_attn_fwd_ws(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(lse), B, Tn, H, Dh, scale, causal, pdrop, 99, None,
                                             C.io_dtype(qkv), C.ptr(wsf) if nf else None, nf, C.stream_ptr()), "fwd")
    nb = C.lib().tsasr_relpos_attn_bwd_workspace_bytes(B, Tn, H)
    res = []
    for fill in (0xFF, 0x00):
        ws = torch.full((nb,), fill, dtype=torch.uint8, device=DEV)
        dqkv, dpk = torch.empty_like(qkv), torch.empty_like(pk)
        du, dv = torch.empty_like(u), torch.empty_like(v)
        C.check(C.lib().tsasr_relpos_attn_bwd(C.ptr(qkv), C.ptr(pk), C.ptr(u), C.ptr(v), C.ptr(lens), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(dpk),
                                              C.ptr(du), C.ptr(dv), B, Tn, H, Dh, scale, causal, pdrop, 99, None, C.io_dtype(qkv), C.ptr(ws), nb, C.stream_ptr()), "bwd")
        torch.cuda.synchronize()
        res.append([t.float().cpu() for t in (dqkv, dpk, du, dv)])
    for a, b_, name in zip(res[0], res[1], ("dqkv", "dpk", "du", "dv")):
        assert torch.isfinite(a).all(), name
        assert torch.equal(a, b_), name
