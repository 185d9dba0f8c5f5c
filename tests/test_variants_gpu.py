"""GPU parity of (a) the BENCHMARKED bf16 HIP path - HIP GEMMs, fused FFN epilogues, persistent LSTM, block-2 front-end GEMM - stage by
stage against the reference's golden vectors and every parameter gradient against the oracle (round 1 pinned only the fp32 mode, whose
GEMMs / LSTM are library calls), and (b) BASELINE.json configs[3], the pretrained-speaker variant (train_librispeechmix_pretrained.py),
in both compute dtypes. Budgets are relative L2 errors ||x - ref|| / ||ref||; bf16 stores round to 2^-9 relative per tensor."""
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
from oracle.golden_recipe import CFG1, SPEAKER_EMBEDDING_DIM, det_tensor, golden_enroll_emb, golden_inputs  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def c2_lattice_mask(shape):
    """oracle/gen_golden_d256.py lattice_mask: 1 inside each utterance's RNN-T lattice, 0 outside."""
    from oracle.golden_recipe import CFG2 as c
    B, Tp, U1, _ = shape
    m = torch.zeros(B, Tp, U1, 1)
    for b in range(B):
        tb, ub = int(round(float(c["mix_lens"][b]) * Tp)), int(round(float(c["tok_lens"][b]) * (U1 - 1)))
        m[b, :tb, : ub + 1] = 1.0
    return m


def rel_l2(a, b):
    b = T(np.asarray(b)) if not isinstance(b, torch.Tensor) else b.detach().float().cpu()
    return float((a.detach().float().cpu() - b).norm() / (b.norm() + 1e-30))


def make_batch(inp, emb=None):
    bm = importlib.import_module("ts-asr_amd.batch")
    f = {
        "id": ["a", "b", "c", "d"],
        "mixed_sig": bm.PaddedData(T(inp["mixed_sig"]), T(inp["mixed_lens"])),
        "enroll_sig": bm.PaddedData(T(inp["enroll_sig"]), T(inp["enroll_lens"])),
        "tokens_bos": bm.PaddedData(T(inp["tokens_bos"]), T(inp["tokens_bos_lens"])),
        "tokens": bm.PaddedData(T(inp["tokens"]), T(inp["tokens_lens"])),
    }
    if emb is not None:
        f["enroll_emb"] = bm.PaddedData(T(emb), T(inp["enroll_lens"]))
    return bm.PaddedBatch(f)


def state_dict_cpu(brain, grad=False):
    return {f"{n}.{k}": v.detach().cpu().float().clone().requires_grad_(grad and v.dtype.is_floating_point)
            for n, m in brain.modules.items() for k, v in m.state_dict().items()}


# per-stage budgets of the bf16 path (measured x ~2.5; the fp32-mode values of tests/test_model_gpu.py are 6e-3 for everything behind
# an MFMA contraction): features are fp32 kernels whose output is rounded once; every later stage stacks GEMM + row kernels in bf16
# (measured on MI355X: frontend 4.7e-3, spk_enc 8.5e-3, enc 7.2e-3, enc_proj 7.5e-3, dec 2.1e-3, dec_proj 2.5e-3, logits 9.2e-3)
BF16_BUDGET = {"frontend": 1.2e-2, "spk_enc": 2e-2, "enc": 2e-2, "enc_proj": 2e-2, "dec": 6e-3, "dec_proj": 7e-3, "logits": 2.5e-2}


def test_every_stage_bf16_vs_reference_golden(golden):
    """bf16 twin of tests/test_model_gpu.py::test_every_stage_fp32_vs_reference_golden: the path bench.py times."""
    brain, h = entry._config1_brain(DEV, "bf16")
    brain._setup_dtype()
    brain.modules.eval()
    g, gf = golden["c1_chain_cat"], golden["c1_features"]
    inp = golden_inputs()
    m = brain.modules
    dev = lambda k: T(inp[k]).to(DEV)  # noqa: E731
    core = importlib.import_module("ts-asr_amd.core")
    seen = {}
    with torch.no_grad():
        fe = m.frontend(T(gf["norm"]).to(DEV))
        assert fe.dtype == torch.bfloat16
        seen["frontend"] = rel_l2(fe[[0, 3]], g["frontend_b03"])
        se = m.speaker_encoder(m.speaker_frontend(T(gf["spk_norm"]).to(DEV)), dev("enroll_lens"))
        seen["spk_enc"] = rel_l2(se, g["spk_enc"])
        enc = m.encoder(fe, dev("mixed_lens"), T(g["spk_emb"]).to(DEV), dev("enroll_lens"))
        seen["enc"] = rel_l2(enc, g["enc"])
        seen["enc_proj"] = rel_l2(m.encoder_proj(enc), g["enc_proj"])
        d, _ = m.decoder(m.embedding(dev("tokens_bos")), lengths=dev("tokens_bos_lens"))     # persistent HIP LSTM (bf16 only)
        seen["dec"] = rel_l2(d, g["dec"])
        seen["dec_proj"] = rel_l2(m.decoder_proj(d), g["dec_proj"])
        logits, hyps = brain.compute_forward(make_batch(inp), core.Stage.VALID)
        seen["logits"] = rel_l2(logits, g["logits"])
    print("bf16 stage errors:", {k: round(v, 5) for k, v in seen.items()})
    for k, v in seen.items():
        assert v < BF16_BUDGET[k], (k, v)
    exact = sum(hyps[b] == g["greedy_hyps"][b, : g["greedy_lens"][b]].tolist() for b in range(4))
    assert exact >= 3, hyps     # token alignments: bit-exact in fp32 mode (test_model_gpu.py); bf16 logits may flip a near-tie


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_training_gradients_vs_oracle(dtype):
    """Every parameter gradient of the mean RNN-T loss, benchmarked bf16 path (and fp32 mode) vs oracle autograd on the same weights.
    Budget: relative L2 per parameter; bf16 activations + bf16 MFMA operands through 2 encoder layers, LSTM and joint."""
    budget = 6e-2 if dtype == "bf16" else 2e-4     # fp32 mode = exact fp32 arithmetic everywhere (round 4): measured worst 2.1e-5 (pos_bias_v)
    brain, h = entry._config1_brain(DEV, dtype)
    brain.modules.train()  # dropout = 0 in this config
    brain.on_fit_start()
    inp = golden_inputs()
    core = importlib.import_module("ts-asr_amd.core")
    sd = state_dict_cpu(brain, grad=True)
    brain.arena.begin_backward(False)
    batch = make_batch(inp)
    out = brain.compute_forward(batch, core.Stage.TRAIN)
    loss = brain.compute_objectives(out, batch, core.Stage.TRAIN)
    loss.backward()
    for s in brain._aux_streams:
        torch.cuda.current_stream().wait_stream(s)
    brain.arena.finish_backward()
    logits_o = R.compute_forward({k: T(v) for k, v in inp.items()}, sd, CFG1, "cat")
    loss_o = RR.transducer_loss_ref_torch(logits_o, T(inp["tokens"]), T(inp["mixed_lens"]), T(inp["tokens_lens"]), 0, "mean")
    loss_o.backward()
    assert float(loss) == pytest.approx(float(loss_o), rel=3e-2 if dtype == "bf16" else 2e-5)
    worst, n, bad = ("", 0.0), 0, []
    for mn, mod in brain.modules.items():
        for k, p in mod.named_parameters():
            if not p.requires_grad:
                continue
            ref = sd[f"{mn}.{k}"].grad
            rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
            if rel > worst[1]:
                worst = (f"{mn}.{k}", rel)
            # the positional path (pos_bias_u / pos_bias_v / linear_pos) sums strongly cancelling terms over every (b, i, j): its gradients'
            # norms are small against the per-term bf16 rounding of dS, so their relative error is the largest of all parameters
            # (round 5, measured: linear_pos.weight 7.0e-2 the worst, every other parameter < 6e-2: the query-major backward still hands dS to the
            # key-major / d(pk) passes in bf16 - the one-kernel backward that would keep it in fp32 was slower (csrc/lab/attention_fused.hip) - so
            # 8e-2 here instead of round 4's 1.5e-1. fp32 mode: no extra allowance - 2.4e-5 is the worst of all)
            lim = 8e-2 if (dtype == "bf16" and ("pos_bias" in k or "linear_pos" in k)) else budget
            if rel >= lim:
                bad.append((mn, k, rel))
            n += 1
    print(f"{dtype}: worst relative L2 gradient error {worst} over {n} parameters")
    assert not bad, bad
    assert n > 150


# ------------------------------------------------------------------------- full layer widths against the reference itself
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_full_width_model_vs_reference_golden(golden, dtype):
    """The SHIPPED kernel widths (Dh = 64 attention, J = 640 joint on the packed-fp16 / f16-MFMA path, H = 512 persistent LSTM, d_ffn 2048
    GEMMs), whole model, against the reference's own outputs and gradients (tests/golden/c2_fullwidth.npz: 2 + 2 layers, B = 2, made by
    oracle/gen_golden_d256.py from the imported reference) - not only against the oracle. Stages, greedy alignments, and the gradient of a
    fixed linear probe on the logits for every parameter (norms for all, values for the small ones)."""
    from oracle.golden_recipe import CFG2, det_tensor
    c = CFG2
    brain, h = entry._config1_brain(DEV, dtype, d_model=c["d_model"], nhead=c["nhead"], encoder_num_layers=c["encoder_num_layers"],
                                    speaker_num_layers=c["speaker_num_layers"], d_ffn=c["d_ffn"], joint_dim=c["joint_dim"],
                                    decoder_neurons=c["decoder_neurons"])
    g = golden["c2_fullwidth"]
    inp = golden_inputs(c)
    core = importlib.import_module("ts-asr_amd.core")
    bm = importlib.import_module("ts-asr_amd.batch")
    batch = bm.PaddedBatch({
        "id": ["a", "b"],
        "mixed_sig": bm.PaddedData(T(inp["mixed_sig"]), T(inp["mixed_lens"])), "enroll_sig": bm.PaddedData(T(inp["enroll_sig"]), T(inp["enroll_lens"])),
        "tokens_bos": bm.PaddedData(T(inp["tokens_bos"]), T(inp["tokens_bos_lens"])), "tokens": bm.PaddedData(T(inp["tokens"]), T(inp["tokens_lens"]))})
    m = brain.modules
    brain._setup_dtype()
    m.eval()
    dev = lambda k: T(inp[k]).to(DEV)  # noqa: E731
    seen = {}
    with torch.no_grad():
        fe = m.frontend(T(g["norm"]).to(DEV))
        enc = m.encoder(fe, dev("mixed_lens"), T(g["spk_emb"]).to(DEV), dev("enroll_lens"))
        seen["enc"] = rel_l2(enc, g["enc"])
        seen["enc_proj"] = rel_l2(m.encoder_proj(enc), g["enc_proj"])
        d, _ = m.decoder(m.embedding(dev("tokens_bos")), lengths=dev("tokens_bos_lens"))
        seen["dec_proj"] = rel_l2(m.decoder_proj(d), g["dec_proj"])
        logits, hyps = brain.compute_forward(batch, core.Stage.VALID)
        seen["logits"] = rel_l2(logits, g["logits"])
    print(dtype, "full-width stage errors:", {k: round(v, 6) for k, v in seen.items()})
    budget = {"enc": 2e-2, "enc_proj": 2e-2, "dec_proj": 7e-3, "logits": 2.5e-2} if dtype == "bf16" else dict.fromkeys(seen, 2e-5)
    for k, v in seen.items():
        assert v < budget[k], (k, v)
    exact = sum(hyps[b] == g["greedy_hyps"][b, : g["greedy_lens"][b]].tolist() for b in range(c["B"]))
    assert exact == c["B"] if dtype == "fp32" else exact >= c["B"] - 1, hyps
    # ---- gradients of the probe
    m.train()       # dropout = 0 in this config
    brain.on_fit_start()
    brain.arena.begin_backward(False)
    logits, _ = brain.compute_forward(batch, core.Stage.TRAIN)
    probe = (T(det_tensor("probe.logits.c2", tuple(logits.shape), 1.0)) * c2_lattice_mask(tuple(logits.shape))).to(DEV)
    (logits.float() * probe).sum().mul(1.0 / logits.numel()).backward()
    for s_ in brain._aux_streams:
        torch.cuda.current_stream().wait_stream(s_)
    brain.arena.finish_backward()
    # fp32 mode: measured worst 3.1e-6 against the reference (budget 2e-4). bf16: a random-sign linear probe over every logit makes every parameter
    # gradient a sum of strongly cancelling terms (unlike the RNN-T loss of test_training_gradients_vs_oracle, budget 6e-2): measured worst 1.5e-1
    # (speaker branch, the longest chain behind the probe: mean-pool -> injection -> two encoder layers -> joint); budget 2e-1 for all parameters
    lim = 2e-1 if dtype == "bf16" else 2e-4
    n, worst, bad, allv = 0, ("", 0.0), [], []
    for mn, mod in m.items():
        for k, p in mod.named_parameters():
            if not p.requires_grad or f"norm:{mn}.{k}" not in g.files:
                continue
            gn, rn = float(p.grad.double().norm()), float(g[f"norm:{mn}.{k}"])
            rel = abs(gn - rn) / (rn + 1e-12)
            if f"grad:{mn}.{k}" in g.files:      # small parameters: the gradient itself
                ref = T(g[f"grad:{mn}.{k}"])
                rel = max(rel, float((p.grad.cpu().float() - ref).norm() / (ref.norm() + 1e-12)))
            allv.append((rel, f"{mn}.{k}"))
            if rel > worst[1]:
                worst = (f"{mn}.{k}", rel)
            if rel >= lim:
                bad.append((mn, k, rel))
            n += 1
    print(f"{dtype}: worst relative gradient error vs the reference {worst} over {n} parameters; top:", [(round(r, 4), k) for r, k in sorted(allv, reverse=True)[:14]])
    assert not bad, bad
    assert n > 150


# ---------------------------------------------------------------------------------------------- configs[3]
def pretrained_brain(dtype, mode="cat"):
    from oracle.golden_recipe import load_det_weights
    brain, h = entry._config1_brain(DEV, dtype, "conformer-t_wavlm_mi355x.yaml", injection_mode=mode)
    assert brain.variant == "pretrained" and "speaker_encoder" not in brain.modules and "speaker_frontend" not in brain.modules
    assert tuple(brain.modules.speaker_proj.w.weight.shape) == (CFG1["d_model"], SPEAKER_EMBEDDING_DIM)
    return brain, h


@pytest.mark.parametrize("mode", ["cat", "sum", "prod"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pretrained_variant_forward_vs_reference_golden(golden, dtype, mode):
    """train_librispeechmix_pretrained.py: x-vector [B,1,512] -> speaker_proj -> injection -> encoder -> joint, vs tests/golden/c1_pretrained.npz."""
    g = golden["c1_pretrained"]
    brain, h = pretrained_brain(dtype, mode)
    brain._setup_dtype()
    brain.modules.eval()
    inp = golden_inputs()
    core = importlib.import_module("ts-asr_amd.core")
    with torch.no_grad():
        spk = brain.modules.speaker_proj(T(golden_enroll_emb()).to(DEV))
        logits, _ = brain.compute_forward(make_batch(inp, golden_enroll_emb()), core.Stage.VALID)
    e_spk, e_log = rel_l2(spk, g[f"spk_emb:{mode}"]), rel_l2(logits, g[f"logits:{mode}"])
    print(dtype, mode, "spk_emb", e_spk, "logits", e_log)
    assert e_spk < (6e-3 if dtype == "bf16" else 1e-5)
    assert e_log < (3e-2 if dtype == "bf16" else 5e-5)       # fp32 mode: exact fp32 arithmetic (csrc/attention_f32.hip, joint_f32.hip)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pretrained_variant_training_step_vs_oracle(dtype):
    """One fit_batch of the pretrained variant: loss vs the oracle's, gradients of every parameter vs oracle autograd, weights move."""
    brain, h = pretrained_brain(dtype)
    brain.modules.train()
    brain.on_fit_start()
    inp, emb = golden_inputs(), golden_enroll_emb()
    core = importlib.import_module("ts-asr_amd.core")
    sd = state_dict_cpu(brain, grad=True)
    assert not any(k.startswith(("speaker_encoder.", "speaker_frontend.")) for k in sd)
    brain.arena.begin_backward(False)
    batch = make_batch(inp, emb)
    out = brain.compute_forward(batch, core.Stage.TRAIN)
    loss = brain.compute_objectives(out, batch, core.Stage.TRAIN)
    loss.backward()
    for s in brain._aux_streams:
        torch.cuda.current_stream().wait_stream(s)
    brain.arena.finish_backward()
    ob = {k: T(v) for k, v in inp.items()}
    ob["enroll_emb"] = T(emb)
    logits_o = R.compute_forward(ob, sd, CFG1, "cat")
    loss_o = RR.transducer_loss_ref_torch(logits_o, T(inp["tokens"]), T(inp["mixed_lens"]), T(inp["tokens_lens"]), 0, "mean")
    loss_o.backward()
    assert float(loss) == pytest.approx(float(loss_o), rel=3e-2 if dtype == "bf16" else 2e-5)
    worst = 0.0
    for mn, mod in brain.modules.items():
        for k, p in mod.named_parameters():
            if p.requires_grad:
                ref = sd[f"{mn}.{k}"].grad
                rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
                worst = max(worst, rel)
                lim = 2e-4 if dtype == "fp32" else 6e-2   # every parameter, the positional ones included (bf16: measured worst 5.6e-2; fp32: 1.3e-5)
                assert rel < lim, (mn, k, rel)
    print(dtype, "pretrained variant: worst relative L2 gradient error", worst)
    w0 = brain.modules.speaker_proj.w.weight.detach().clone()
    brain.arena.zero_()
    brain.fit_batch(make_batch(inp, emb))
    assert not torch.equal(w0, brain.modules.speaker_proj.w.weight) and brain.flush_nonfinite() == 0


# ---------------------------------------------------------------------------------------------- the recipe without a speaker encoder
def none_brain(dtype):
    brain, h = entry._config1_brain(DEV, dtype, "conformer-t_none_mi355x.yaml")
    assert brain.variant == "none" and not any(k.startswith("speaker") for k in brain.modules)
    return brain, h


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_none_variant_forward_vs_reference_golden(golden, dtype):
    """train_librispeechmix_none.py (no speaker branch, encoder(feats, lens)) on the HIP path vs tests/golden/c1_none.npz."""
    g = golden["c1_none"]
    brain, h = none_brain(dtype)
    brain._setup_dtype()
    brain.modules.eval()
    core = importlib.import_module("ts-asr_amd.core")
    with torch.no_grad():
        logits, _ = brain.compute_forward(make_batch(golden_inputs()), core.Stage.VALID)
    e_log = rel_l2(logits, g["logits:full"])
    print(dtype, "none variant logits", e_log)
    assert e_log < (3e-2 if dtype == "bf16" else 5e-5)       # fp32 mode: exact fp32 arithmetic (csrc/attention_f32.hip, joint_f32.hip)


def test_none_variant_training_steps_and_graph_replay():
    """fit_batch of the `none` recipe in bf16: the loss of the first step equals the oracle's (no speaker keys in its state dict), every
    parameter gradient matches the oracle's autograd, and hipGraph replays reproduce the eager losses bit for bit (this variant has no
    forked stream at all: one chain of launches)."""
    inp = golden_inputs()
    core = importlib.import_module("ts-asr_amd.core")
    brain, h = none_brain("bf16")
    brain.modules.train()
    brain.on_fit_start()
    sd = state_dict_cpu(brain, grad=True)
    assert not any(k.startswith("speaker") for k in sd)
    brain.arena.begin_backward(False)
    batch = make_batch(inp)
    out = brain.compute_forward(batch, core.Stage.TRAIN)
    loss = brain.compute_objectives(out, batch, core.Stage.TRAIN)
    loss.backward()
    brain.arena.finish_backward()
    ob = {k: T(v) for k, v in inp.items() if not k.startswith("enroll")}
    logits_o = R.compute_forward(ob, sd, CFG1, None)
    loss_o = RR.transducer_loss_ref_torch(logits_o, T(inp["tokens"]), T(inp["mixed_lens"]), T(inp["tokens_lens"]), 0, "mean")
    loss_o.backward()
    assert float(loss) == pytest.approx(float(loss_o), rel=3e-2)
    worst_none = 0.0
    for mn, mod in brain.modules.items():
        for k, p in mod.named_parameters():
            if p.requires_grad:
                ref = sd[f"{mn}.{k}"].grad
                rel = float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12))
                worst_none = max(worst_none, rel)
                assert rel < 6e-2, (mn, k, rel)      # every parameter, the positional ones included (measured worst 5.5e-2)
    print("bf16 none variant: worst relative L2 gradient error", worst_none)
    losses = {}
    for mode in ("eager", "graph"):
        b2, _ = none_brain("bf16")
        b2.modules.train()
        if mode == "graph":
            b2.enable_hip_graph(warmup_steps=2)
        bt = make_batch(inp).to(DEV)
        losses[mode] = [float(b2.fit_batch(bt)) for _ in range(6)]
        if mode == "graph":
            assert len(b2._graphs) == 1
    assert losses["eager"] == losses["graph"], losses
    assert losses["eager"][-1] < losses["eager"][0]


# ---------------------------------------------------------------------------------------------- configs[1]: the whole model at full size
def test_config1_full_size_whole_model_properties():
    """BASELINE.json configs[1] (the benchmarked workload: 12 + 6 layers, d_model 256, B = 32, T = 1000 mel frames, U = 120, bf16, dropout
    on, ragged lengths) through TSASR.compute_forward / compute_objectives / backward at FULL size - where no fixture fits, the properties
    the domain offers: the loss equals the C oracle's (torchaudio semantics restated in float64: oracle/rnnt_ref.c) on the logits the HIP
    path produced; every row of dlogits sums to 0 over the vocabulary and is 0 outside each utterance's lattice; every parameter gets a
    finite gradient; a second, fresh run of two whole training steps is bit-identical (losses and every weight)."""
    import bench
    from oracle import rnnt_ref as RRc
    core = importlib.import_module("ts-asr_amd.core")
    bm = importlib.import_module("ts-asr_amd.batch")
    B, T, U = 32, 1000, 120

    def fresh():
        torch.manual_seed(0)
        importlib.import_module("ts-asr_amd.ops").seed_state(torch.device(DEV)).zero_()   # the device-side dropout step counter is per process
        brain, h, _ = bench.build_brain(DEV, "bf16", 1)
        batch = bm.synthetic_batch(B, T, 500, U, feats=True, seed=1234, ragged=True).to(DEV)
        return brain, batch

    brain, batch = fresh()
    brain.on_fit_start()
    importlib.import_module("ts-asr_amd.ops").begin_step(DEV)
    brain.arena.begin_backward(False)
    logits, _ = brain.compute_forward(batch, core.Stage.TRAIN)
    assert tuple(logits.shape) == (B, T // 4, U + 1, 29)
    grabbed = []
    logits.register_hook(lambda g: grabbed.append(g.detach().clone()))
    loss = brain.compute_objectives((logits, None), batch, core.Stage.TRAIN)
    loss.backward()
    for s in brain._aux_streams:
        torch.cuda.current_stream().wait_stream(s)
    brain.arena.finish_backward()
    torch.cuda.synchronize()
    tl = (batch.mixed_sig.lengths.cpu() * (T // 4)).round().int().numpy()
    ul = (batch.tokens.lengths.cpu() * U).round().int().numpy()
    costs_ref, _ = RRc.rnnt_costs_grads(np.ascontiguousarray(logits.detach().float().cpu().numpy()), batch.tokens.data.cpu().int().numpy(), tl, ul, 0, want_grads=False)
    assert float(loss) == pytest.approx(float(costs_ref.mean()), rel=2e-5)
    dl = grabbed[0].float()
    assert float(dl.sum(-1).abs().max()) < 5e-6
    tmask = torch.arange(T // 4, device=DEV)[None, :, None] >= torch.as_tensor(tl, device=DEV)[:, None, None]
    umask = torch.arange(U + 1, device=DEV)[None, None, :] > torch.as_tensor(ul, device=DEV)[:, None, None]
    assert float(dl[(tmask | umask).expand(B, T // 4, U + 1)].abs().max()) == 0.0
    n = 0
    for mn, mod in brain.modules.items():
        for k, p in mod.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), (mn, k)
                n += 1
    gn = float(brain.arena.grads.norm())
    assert n > 400 and np.isfinite(gn) and gn > 0
    del brain, logits, loss, dl, grabbed
    runs = []
    for _ in range(2):
        b2, bt = fresh()
        b2.modules.train()
        ls = [float(b2.fit_batch(bt)) for _ in range(2)]
        torch.cuda.synchronize()
        runs.append((ls, b2.arena.flat_params.detach().clone()))
        del b2
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][1] != runs[0][0][0] and all(np.isfinite(runs[0][0]))


def test_deferred_grouped_dpk_is_bit_identical_to_per_layer_launches(monkeypatch):
    """The d(pk) passes of the attention backward are queued during the arena's backward and run as ONE grouped launch in front of the
    grouped weight gradients (csrc/attention.hip, tsasr_relpos_dpk_defer / _flush; ops.dpk_flush): same blocks, same arithmetic, same order
    of sums as the per-layer launches - every gradient of the step, linear_pos.weight's included, must be bit-identical with the knob
    off; and the deferred path must really have been taken (2 + 2 layers queued at configs[0])."""
    ops = importlib.import_module("ts-asr_amd.ops")
    capi = importlib.import_module("ts-asr_amd._capi")
    core = importlib.import_module("ts-asr_amd.core")
    inp = golden_inputs()
    grads, seen = {}, {}
    real_flush = ops.dpk_flush
    for flag in (True, False):
        monkeypatch.setattr(ops, "_DPK_DEFER", flag)
        pend = []
        monkeypatch.setattr(ops, "dpk_flush", lambda pend=pend: (pend.append(capi.lib().tsasr_relpos_dpk_pending()), real_flush())[1])
        brain, h = entry._config1_brain(DEV, "bf16")
        brain.modules.train()
        brain.on_fit_start()
        ops.begin_step(DEV)
        brain.arena.begin_backward(False)
        batch = make_batch(inp)
        out = brain.compute_forward(batch, core.Stage.TRAIN)
        loss = brain.compute_objectives(out, batch, core.Stage.TRAIN)
        loss.backward()
        for s_ in brain._aux_streams:
            torch.cuda.current_stream().wait_stream(s_)
        brain.arena.finish_backward()
        torch.cuda.synchronize()
        grads[flag] = {f"{mn}.{k}": p.grad.detach().clone() for mn, mod in brain.modules.items() for k, p in mod.named_parameters() if p.requires_grad}
        seen[flag] = sum(pend)
        assert capi.lib().tsasr_relpos_dpk_pending() == 0
    assert seen[True] == 4 and seen[False] == 0, seen          # 2 mixture + 2 speaker layers went through the grouped launch
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k
    assert float(grads[True]["encoder.layers.0.mha_layer.linear_pos.weight"].abs().max()) > 0
