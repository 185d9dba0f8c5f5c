"""Pins the CPU oracle (oracle/tsasr_ref.py) against golden vectors produced by the reference itself
(oracle/gen_golden.py imported /root/reference in the build container). fp32, tolerance stated per test."""
import numpy as np
import pytest
import torch

from oracle import tsasr_ref as R
from oracle.golden_recipe import CFG1, CFG2, det_tensor, det_weight, golden_inputs

ATOL = 2e-5  # fp32 forward tolerance on O(1) activations (1e-5 .. a few ulp of sums)


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


def close(a, b, atol=ATOL, rtol=1e-4):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def sd_from_shapes(prefix, shapes):
    """state_dict with deterministic weights for the given {key: shape} (prefix is part of the name)."""
    out = {}
    for k, s in shapes.items():
        w = det_weight(prefix + k, s)
        out[k] = None if w is None else T(w)
    return out


def frontend_shapes():
    s = {}
    for i, (cin, F) in enumerate(((1, 40), (128, 20))):
        p = f"convblock_{i}."
        s[p + "convs.conv_0.conv.weight"] = (128, cin, 3, 3)
        s[p + "convs.conv_0.conv.bias"] = (128,)
        s[p + "convs.norm_0.norm.weight"] = (F, 128)
        s[p + "convs.norm_0.norm.bias"] = (F, 128)
        s[p + "reduce_conv.conv.conv.weight"] = (128, cin, 1, 1)
        s[p + "reduce_conv.conv.conv.bias"] = (128,)
        s[p + "reduce_conv.norm.norm.weight"] = (F, 128)
        s[p + "reduce_conv.norm.norm.bias"] = (F, 128)
    return s


def layer_shapes(p, D, F, K=31, H=4):
    Dh = D // H
    s = {
        p + "mha_layer.in_proj_weight": (3 * D, D), p + "mha_layer.pos_bias_u": (Dh, H), p + "mha_layer.pos_bias_v": (Dh, H),
        p + "mha_layer.out_proj.weight": (D, D), p + "mha_layer.out_proj.bias": (D,), p + "mha_layer.linear_pos.weight": (D, D),
        p + "convolution_module.layer_norm.weight": (D,), p + "convolution_module.layer_norm.bias": (D,),
        p + "convolution_module.bottleneck.0.weight": (2 * D, D, 1), p + "convolution_module.bottleneck.0.bias": (2 * D,),
        p + "convolution_module.conv.weight": (D, 1, K), p + "convolution_module.conv.bias": (D,),
        p + "convolution_module.after_conv.0.weight": (D,), p + "convolution_module.after_conv.0.bias": (D,),
        p + "convolution_module.after_conv.2.weight": (D, D), p + "convolution_module.after_conv.2.bias": (D,),
        p + "norm1.norm.weight": (D,), p + "norm1.norm.bias": (D,), p + "norm2.norm.weight": (D,), p + "norm2.norm.bias": (D,),
    }
    for f in ("ffn_module1.", "ffn_module2."):
        s[p + f + "0.weight"] = (D,)
        s[p + f + "0.bias"] = (D,)
        s[p + f + "1.ffn.0.weight"] = (F, D)
        s[p + f + "1.ffn.0.bias"] = (F,)
        s[p + f + "1.ffn.3.weight"] = (D, F)
        s[p + f + "1.ffn.3.bias"] = (D,)
    return s


def encoder_shapes(cfg, nl, mode):
    D = cfg["d_model"]
    s = {"custom_src_module.layers.0.w.weight": (D, cfg["encoder_input_size"]), "custom_src_module.layers.0.w.bias": (D,),
         "norm.norm.weight": (D,), "norm.norm.bias": (D,)}
    for i in range(nl):
        s.update(layer_shapes(f"layers.{i}.", D, cfg["d_ffn"], cfg["kernel_size"], cfg["nhead"]))
    if mode == "cat":
        s["cat_proj.w.weight"] = (D, 2 * D)
        s["cat_proj.w.bias"] = (D,)
    elif mode == "cross_attention":
        s.update({"speaker_attn.att.in_proj_weight": (3 * D, D), "speaker_attn.att.in_proj_bias": (3 * D,),
                  "speaker_attn.att.out_proj.weight": (D, D), "speaker_attn.att.out_proj.bias": (D,)})
    return s


def full_state_dict(cfg, mode):
    """Every Brain module's deterministic state_dict under its module name (as gen_golden.build does)."""
    D, J, Hd, V = cfg["d_model"], cfg["joint_dim"], cfg["decoder_neurons"], cfg["vocab_size"]
    shapes = {}
    for k, s in frontend_shapes().items():
        shapes["frontend." + k] = s
        shapes["speaker_frontend." + k] = s
    for k, s in encoder_shapes(cfg, cfg["encoder_num_layers"], mode).items():
        shapes["encoder." + k] = s
    for k, s in encoder_shapes(cfg, cfg["speaker_num_layers"], "prod").items():  # default injection_mode, no extra params
        shapes["speaker_encoder." + k] = s
    shapes.update({
        "encoder_proj.w.weight": (J, D), "encoder_proj.w.bias": (J,),
        "decoder.rnn.weight_ih_l0": (4 * Hd, V - 1), "decoder.rnn.weight_hh_l0": (4 * Hd, Hd),
        "decoder.rnn.bias_ih_l0": (4 * Hd,), "decoder.rnn.bias_hh_l0": (4 * Hd,),
        "decoder_proj.w.weight": (J, Hd), "decoder_proj.w.bias": (J,),
        "transducer_head.w.weight": (V, J), "transducer_head.w.bias": (V,),
        "speaker_proj.w.weight": (D, D), "speaker_proj.w.bias": (D,),
    })
    return {k: T(det_weight(k, s)) for k, s in shapes.items()}


def torch_batch(inp):
    return {k: T(v) for k, v in inp.items()}


# ------------------------------------------------------------------------------------------------
def test_fbank_and_sentence_norm(golden):
    g = golden["c1_features"]
    inp = golden_inputs()
    fb = R.fbank(T(inp["mixed_sig"]))
    close(fb, g["fbank"], atol=2e-3, rtol=1e-4)  # dB scale (values up to ~40 dB), 10*log10 amplifies fp32 FFT noise
    close(R.sentence_norm(T(g["fbank"]), T(inp["mixed_lens"])), g["norm"], atol=1e-4)
    sfb = R.fbank(T(inp["enroll_sig"]))
    close(sfb, g["spk_fbank"], atol=2e-3, rtol=1e-4)
    close(R.sentence_norm(T(g["spk_fbank"]), T(inp["enroll_lens"])), g["spk_norm"], atol=1e-4)


def test_fbank_reference_known_answers():
    """vendor/speechbrain/tests/unittests/test_features.py:57-84: zeros -> -100 dB; top_db clamp."""
    mel = torch.zeros(3, 11, 257) @ R.mel_filterbank()
    x_db = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
    assert torch.equal(x_db, torch.full_like(x_db, -100.0))


def test_sentence_norm_reference_known_answer():
    """vendor/speechbrain/tests/unittests/test_features.py:94-113."""
    x = torch.tensor([1.0, 2, 3, 0, 0, 0]).view(1, 6, 1)
    out = R.sentence_norm(x, torch.tensor([0.5])).squeeze()
    assert torch.equal(out, torch.tensor([-1.0, 0, 1, -2, -2, -2]))


def test_frontend(golden):
    g, gv = golden["c1_chain_cat"], golden["c1_encoder_variants"]
    norm = T(golden["c1_features"]["norm"])
    sd = sd_from_shapes("frontend.", frontend_shapes())
    out = R.frontend(norm, sd, "same")
    assert out.shape == (4, 50, 20, 128)
    close(out[[0, 3]], g["frontend_b03"], atol=1e-4)
    close(R.frontend(norm, sd, "causal")[0], gv["frontend_causal_b0"], atol=1e-4)


def test_relpos_table(golden):
    close(R.relpos_table(50, 144), golden["c1_blocks"]["relpos_table"], atol=1e-6)


def _block_inputs(golden):
    g = golden["c1_blocks"]
    x = T(g["x"]).clone().requires_grad_(True)
    lens = T(np.asarray(CFG1["mix_lens"], np.float32))
    kpm = ~R.length_to_mask((lens * 50).round(), 50)
    probe = T(det_tensor("probe.blk", (4, 50, 144), 1.0))
    return g, x, kpm, probe


def _grad_sd(prefix, shapes):
    sd = sd_from_shapes(prefix, shapes)
    for v in sd.values():
        v.requires_grad_(True)
    return sd


@pytest.mark.parametrize("tag,use_kpm,causal", [("mha_nomask", False, False), ("mha_kpm", True, False), ("mha_kpm_causal", True, True)])
def test_relpos_mha(golden, tag, use_kpm, causal):
    g, x, kpm, probe = _block_inputs(golden)
    shapes = {k[len("layers.0.mha_layer."):]: s for k, s in layer_shapes("layers.0.", 144, 576).items() if "mha_layer" in k}
    sd = _grad_sd("blk.mha.", shapes)
    pe = R.relpos_table(50, 144)
    out, attn = R.relpos_mha(x, pe, sd, "", 4, kpm if use_kpm else None, causal, return_attn=True)
    (out * probe).sum().backward()
    close(out, g[f"{tag}:out"])
    close(x.grad, g[f"{tag}:dx"], atol=1e-4)
    for k in ("in_proj_weight", "pos_bias_u", "pos_bias_v", "linear_pos.weight", "out_proj.weight", "out_proj.bias"):
        close(sd[k].grad, g[f"{tag}:d.{k}"], atol=2e-4, rtol=1e-3)
    if tag == "mha_kpm":
        close(attn[1], g["mha_kpm:attn_b1"], atol=1e-6)


@pytest.mark.parametrize("tag,causal", [("conv", False), ("conv_causal", True)])
def test_conv_module(golden, tag, causal):
    g, x, kpm, probe = _block_inputs(golden)
    shapes = {k[len("layers.0.convolution_module."):]: s for k, s in layer_shapes("layers.0.", 144, 576).items() if "convolution_module" in k}
    sd = _grad_sd("blk.conv.", shapes)
    out = R.conv_module(x, sd, "", kpm, causal)
    (out * probe).sum().backward()
    close(out, g[f"{tag}:out"])
    close(x.grad, g[f"{tag}:dx"], atol=1e-4)
    for k in shapes:
        close(sd[k].grad, g[f"{tag}:d.{k}"], atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("tag,causal", [("layer", False), ("layer_causal", True)])
def test_conformer_layer(golden, tag, causal):
    g, x, kpm, probe = _block_inputs(golden)
    shapes = {k[len("layers.0."):]: s for k, s in layer_shapes("layers.0.", 144, 576).items()}
    sd = _grad_sd("blk.layer.", shapes)
    out = R.conformer_layer(x, R.relpos_table(50, 144), sd, "", 4, kpm, causal)
    (out * probe).sum().backward()
    close(out, g[f"{tag}:out"], atol=5e-5)
    close(x.grad, g[f"{tag}:dx"], atol=2e-4, rtol=1e-3)
    for k in g.files:
        if k.startswith(tag + ":d."):
            close(sd[k[len(tag) + 3:]].grad, g[k], atol=5e-4, rtol=2e-3)
    # ffn sub-block
    x2 = T(g["x"]).clone().requires_grad_(True)
    sd2 = _grad_sd("blk.layer.", shapes)
    y = R.ffn_module(x2, sd2, "ffn_module1.")
    (y * probe).sum().backward()
    if tag == "layer":
        close(y, g["ffn:out"])
        close(x2.grad, g["ffn:dx"], atol=1e-4)
        close(sd2["ffn_module1.1.ffn.0.weight"].grad, g["ffn:d.ffn_module1.1.ffn.0.weight"], atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("mode", ["cat", "sum", "prod", "cross_attention"])
@pytest.mark.parametrize("causal", [False, True])
def test_full_chain_variants(golden, mode, causal):
    gv = golden["c1_encoder_variants"]
    sd = full_state_dict(CFG1, mode)
    batch = torch_batch(golden_inputs())
    with torch.no_grad():
        c = {}
        logits = R.compute_forward(batch, sd, CFG1, mode, causal, "causal" if causal else "same", collect=c)
    tag = f"{mode}{'_causal' if causal else ''}"
    close(c["enc"], gv["enc:" + tag], atol=2e-4, rtol=1e-3)
    close(logits[1], gv["logits_b1:" + tag], atol=2e-4, rtol=1e-3)


def test_full_chain_cat_every_stage_and_greedy(golden):
    g = golden["c1_chain_cat"]
    sd = full_state_dict(CFG1, "cat")
    batch = torch_batch(golden_inputs())
    with torch.no_grad():
        c = {}
        logits = R.compute_forward(batch, sd, CFG1, "cat", collect=c)
        hyps = R.greedy_decode(c["enc_proj"], sd, CFG1)
    for k in ("spk_enc", "spk_pool", "spk_emb", "enc", "enc_proj", "dec", "dec_proj", "logits"):
        close(c[k], g[k], atol=2e-4, rtol=1e-3)
    for b in range(4):  # bit-exact token sequences
        assert hyps[b] == g["greedy_hyps"][b, : g["greedy_lens"][b]].tolist()


def test_full_chain_backward(golden):
    g = golden["c1_chain_cat_grads"]
    sd = full_state_dict(CFG1, "cat")
    for v in sd.values():
        v.requires_grad_(True)
    logits = R.compute_forward(torch_batch(golden_inputs()), sd, CFG1, "cat")
    probe = T(det_tensor("probe.logits", logits.shape, 1.0))
    (logits * probe).sum().mul(1.0 / logits.numel()).backward()
    checked = 0
    for k in g.files:
        kind, name = k.split(":", 1)
        if name not in sd:
            continue
        if kind == "norm":
            np.testing.assert_allclose(sd[name].grad.double().norm().item(), float(g[k]), rtol=2e-3, atol=1e-7)
        else:
            close(sd[name].grad, g[k], atol=2e-5, rtol=5e-3)
        checked += 1
    assert checked > 150

def test_full_width_chain_vs_reference_golden(golden):
    """The oracle at the benchmark's layer widths (oracle/golden_recipe.CFG2: d_model 256, Dh 64, d_ffn 2048, joint 640, predictor 512; 2 + 2
    layers, B = 2) against the reference's own outputs and gradients (oracle/gen_golden_d256.py -> tests/golden/c2_fullwidth.npz)."""
    g = golden["c2_fullwidth"]
    sd = full_state_dict(CFG2, "cat")
    batch = torch_batch(golden_inputs(CFG2))
    with torch.no_grad():
        c = {}
        R.compute_forward(batch, sd, CFG2, "cat", collect=c)
        hyps = R.greedy_decode(c["enc_proj"], sd, CFG2)
    for k in ("norm", "spk_norm", "spk_emb", "enc", "enc_proj", "dec_proj", "logits"):
        close(c[k], g[k], atol=2e-4, rtol=1e-3)
    for b in range(CFG2["B"]):
        assert hyps[b] == g["greedy_hyps"][b, : g["greedy_lens"][b]].tolist()
    for v in sd.values():
        v.requires_grad_(True)
    logits = R.compute_forward(batch, sd, CFG2, "cat")
    probe = T(det_tensor("probe.logits.c2", logits.shape, 1.0)) * c2_lattice_mask(logits.shape)
    (logits * probe).sum().mul(1.0 / logits.numel()).backward()
    checked = 0
    for k in g.files:
        if ":" not in k:
            continue
        kind, name = k.split(":", 1)
        if name not in sd:
            continue
        if kind == "norm":
            np.testing.assert_allclose(sd[name].grad.double().norm().item(), float(g[k]), rtol=2e-3, atol=1e-7)
        else:
            close(sd[name].grad, g[k], atol=2e-5, rtol=5e-3)
        checked += 1
    assert checked > 150


@pytest.mark.parametrize("beam", [4, 15])
def test_beam_search_vs_reference(golden, beam):
    """oracle.beam_decode against the reference's TransducerBeamSearcher (tests/golden/c1_beam.npz, oracle/gen_golden_beam.py):
    token sequences bit-exact, length-normalised scores to 1e-4. The fixture raised the head's blank bias (see the generator)."""
    g = golden["c1_beam"]
    sd = full_state_dict(CFG1, "cat")
    sd["transducer_head.w.bias"] = sd["transducer_head.w.bias"].clone()
    sd["transducer_head.w.bias"][0] += float(g["blank_bias"])
    enc_proj = T(golden["c1_chain_cat"]["enc_proj"])
    with torch.no_grad():
        hyps, scores = R.beam_decode(enc_proj, sd, CFG1, beam_size=beam)
    for b in range(4):
        assert hyps[b] == g[f"beam{beam}_hyps"][b, : g[f"beam{beam}_lens"][b]].tolist(), b
        assert abs(scores[b] - g[f"beam{beam}_scores"][b]) < 1e-4


SA_CASES = {"recipe": dict(replace_with_zero=False), "zero": dict(replace_with_zero=True), "nowarp": dict(replace_with_zero=False)}


def sa_draws(g, name, rep, B=4):
    k = f"sa_{name}_{rep}"
    warp = k + "_c" in g.files
    return dict(c=int(g[k + "_c"][0]) if warp else None, w=int(g[k + "_w"][0]) if warp else None,
                flen=g[k + "_flen"].reshape(B, -1), fpos=g[k + "_fpos"].reshape(B, -1),
                tlen=g[k + "_tlen"].reshape(B, -1), tpos=g[k + "_tpos"].reshape(B, -1)), g[k + "_y"]


@pytest.mark.parametrize("name", list(SA_CASES))
@pytest.mark.parametrize("rep", [0, 1, 2])
def test_spec_augment_vs_reference(golden, name, rep):
    """oracle.spec_augment with the reference's own draws (recorded by oracle/gen_golden_aug.py) against the reference's output:
    recipe settings (fill with the mean), replace_with_zero, and masks only. 1e-5: cubic taps are summed in a different order."""
    g = golden["c1_augment"]
    draws, y = sa_draws(g, name, rep)
    close(R.spec_augment(T(g["sa_x"]), **draws, **SA_CASES[name]), y, atol=1e-5)


def test_spec_augment_skips_warp_on_short_inputs(golden):
    g = golden["c1_augment"]          # time - window <= window (SB/lobes/augment.py:131-132)
    close(R.spec_augment(T(g["sa_short_x"]), c=5, w=6), g["sa_short_y"], atol=0)


@pytest.mark.parametrize("speed", [95, 100, 105, 50])
def test_resample_vs_reference(golden, speed):
    """The polyphase restatement against the reference's Resample (filter bank bit-equal, output to 1e-6), at the recipe's three
    speeds and the reference unit test's half speed (vendor/speechbrain/tests/unittests/test_augment.py:100-113)."""
    g = golden["c1_augment"]
    new = 16000 * speed // 100
    y = R.resample(T(g["sp_x"]), 16000, new)
    assert y.shape[1] == R.resample_out_len(4000, 16000, new) == g[f"sp_{speed}_y"].shape[1]
    close(y, g[f"sp_{speed}_y"], atol=1e-6)
    if speed != 100:
        first, w = R.resample_filters(16000, new)
        assert np.array_equal(first.numpy(), g[f"sp_{speed}_first"]) and np.array_equal(w.numpy(), g[f"sp_{speed}_weights"])
    if speed == 50:
        sine = torch.sin(torch.arange(16000.0)).unsqueeze(0)
        half = R.resample(sine, 16000, 8000)
        close(half, g["sp_sine_half"], atol=2e-6)
        assert half.allclose(sine[:, ::2], atol=3e-1)          # the reference's own assertion


# ---------------------------------------------------------------------------------------------- configs[3]: pretrained-speaker variant
@pytest.mark.parametrize("mode", ["cat", "sum", "prod"])
def test_oracle_pretrained_variant_vs_reference_golden(golden, mode):
    """train_librispeechmix_pretrained.py:45-135 with the frozen encoder's x-vector given (tests/golden/c1_pretrained.npz, generated by
    oracle/gen_golden_pretrained.py from the reference's own modules): speaker_proj(512 -> D), injection, encoder, transducer logits."""
    from oracle.golden_recipe import SPEAKER_EMBEDDING_DIM, golden_enroll_emb
    g = golden["c1_pretrained"]
    inp = golden_inputs()
    sd = full_state_dict(CFG1, mode)
    sd = {k: v for k, v in sd.items() if not k.startswith(("speaker_frontend.", "speaker_encoder."))}
    sd["speaker_proj.w.weight"] = T(det_weight("speaker_proj.w.weight", (CFG1["d_model"], SPEAKER_EMBEDDING_DIM)))
    batch = torch_batch(inp)
    batch["enroll_emb"] = T(golden_enroll_emb())
    if mode == "cat":
        for v in sd.values():
            v.requires_grad_(True)
    c = {}
    with torch.set_grad_enabled(mode == "cat"):
        logits = R.compute_forward(batch, sd, CFG1, mode, collect=c)
    np.testing.assert_allclose(c["spk_emb"].detach().numpy(), g[f"spk_emb:{mode}"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(c["enc"].detach().numpy(), g[f"enc:{mode}"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(logits.detach().numpy(), g[f"logits:{mode}"], atol=2e-4, rtol=1e-4)
    if mode == "cat":   # gradients for the fixed probe: norms of every parameter, values of the small ones
        probe = T(det_tensor("probe.logits", tuple(logits.shape), 1.0))
        (logits * probe).sum().mul(1.0 / logits.numel()).backward()
        n = 0
        for k in g.files:
            if k.startswith("norm:"):
                name = k[5:]
                if name in sd and sd[name].grad is not None:
                    assert float(sd[name].grad.double().norm()) == pytest.approx(float(g[k]), rel=2e-3, abs=1e-7), name
                    n += 1
            elif k.startswith("grad:") and k[5:] in sd:
                np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], atol=2e-6, rtol=2e-3)
        assert n >= 60


# ---------------------------------------------------------------------------------------------- the recipe without a speaker encoder
@pytest.mark.parametrize("tag", ["full", "causal"])
def test_oracle_none_variant_vs_reference_golden(golden, tag):
    """train_librispeechmix_none.py:34-95 (conformer-t_none.yaml: no speaker modules, encoder built without injection arguments and
    called as encoder(feats, lens)) against tests/golden/c1_none.npz (oracle/gen_golden_none.py, the reference's own modules)."""
    g = golden["c1_none"]
    sd = {k: v for k, v in full_state_dict(CFG1, "prod").items() if not k.startswith("speaker_")}
    assert sorted(sd) == [str(k) for k in g["state_keys"] if not str(k).endswith(("inv_freq", "Embedding.weight", "compute_deltas.kernel"))]   # buffers / the frozen one-hot table
    batch = {k: v for k, v in torch_batch(golden_inputs()).items() if not k.startswith("enroll")}
    if tag == "full":
        for v in sd.values():
            v.requires_grad_(True)
    c = {}
    with torch.set_grad_enabled(tag == "full"):
        logits = R.compute_forward(batch, sd, CFG1, None, causal=(tag == "causal"), collect=c)
    np.testing.assert_allclose(c["enc"].detach().numpy(), g[f"enc:{tag}"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(logits.detach().numpy(), g[f"logits:{tag}"], atol=2e-4, rtol=1e-4)
    if tag == "full":
        probe = T(det_tensor("probe.logits", tuple(logits.shape), 1.0))
        (logits * probe).sum().mul(1.0 / logits.numel()).backward()
        n = 0
        for k in g.files:
            if k.startswith("norm:") and k[5:] in sd and sd[k[5:]].grad is not None:
                assert float(sd[k[5:]].grad.double().norm()) == pytest.approx(float(g[k]), rel=2e-3, abs=1e-7), k
                n += 1
        assert n >= 50
