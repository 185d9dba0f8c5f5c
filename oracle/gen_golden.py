#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE; runs ONLY in the build container).

Imports the reference implementation from /root/reference (read-only, never copied), builds the
BASELINE.json configs[0] modules with the constructor arguments of
hparams/LibriSpeechMix/conformer-t_scratch.yaml:121-245, overwrites every parameter with the
deterministic values of ``oracle/golden_recipe.py`` and dumps inputs' and stages' outputs to
``tests/golden/*.npz``. The fixtures are data (inputs + expected outputs); no reference source
travels with them.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden.py

hyperpyyaml / torchaudio / ruamel.yaml are not installed here; the three names are registered as
empty modules before the import - nothing from them is executed on this path.
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)
from golden_recipe import CFG1, det_tensor, golden_inputs, load_det_weights  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("hyperpyyaml", resolve_references=lambda *a, **k: None, load_hyperpyyaml=lambda *a, **k: None)
    ta = _stub("torchaudio")
    ta.functional = _stub("torchaudio.functional")
    ta.transforms = _stub("torchaudio.transforms")
    r = _stub("ruamel")
    r.yaml = _stub("ruamel.yaml")
    sys.path[:0] = ["/root/reference/vendor/speechbrain", "/root/reference"]
    import speechbrain  # noqa: F401


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def N(x):
    return x.detach().cpu().numpy().copy()


def build(cfg, injection_mode="cat", causal=False, frontend_padding="same"):
    from models.conformer import ConformerEncoder
    from speechbrain.lobes.features import Fbank
    from speechbrain.lobes.models.convolution import ConvolutionFrontEnd
    from speechbrain.nnet.embedding import Embedding
    from speechbrain.nnet.linear import Linear
    from speechbrain.nnet.RNN import LSTM
    from speechbrain.nnet.transducer.transducer_joint import Transducer_joint
    from speechbrain.processing.features import InputNormalization

    def fe(padding):
        return ConvolutionFrontEnd(
            input_shape=[None, None, cfg["n_mels"]], num_blocks=2, num_layers_per_block=1,
            out_channels=cfg["frontend_channels"], kernel_sizes=(3, 3), strides=(2, 2),
            residuals=(True, True), dropout=0.0, padding=padding)

    def enc(nl, **kw):
        return ConformerEncoder(
            input_size=cfg["encoder_input_size"], d_model=cfg["d_model"], nhead=cfg["nhead"],
            num_layers=nl, d_ffn=cfg["d_ffn"], dropout=0.0, activation=torch.nn.LeakyReLU,
            kernel_size=cfg["kernel_size"], **kw)

    fb = dict(sample_rate=cfg["sample_rate"], n_fft=cfg["n_fft"], n_mels=cfg["n_mels"], win_length=cfg["win_length"])
    m = dict(
        feature_extractor=Fbank(**fb),
        normalizer=InputNormalization(norm_type="sentence", update_until_epoch=4),
        frontend=fe(frontend_padding),
        encoder=enc(cfg["encoder_num_layers"], causal=causal, injection_mode=injection_mode, injection_after=0),
        encoder_proj=Linear(input_size=cfg["d_model"], n_neurons=cfg["joint_dim"]),
        embedding=Embedding(num_embeddings=cfg["vocab_size"], consider_as_one_hot=True, blank_id=cfg["blank_index"]),
        decoder=LSTM(input_shape=[None, None, cfg["vocab_size"] - 1], hidden_size=cfg["decoder_neurons"], num_layers=1),
        decoder_proj=Linear(input_size=cfg["decoder_neurons"], n_neurons=cfg["joint_dim"]),
        joiner=Transducer_joint(joint="sum", nonlinearity=torch.nn.LeakyReLU),
        transducer_head=Linear(input_size=cfg["joint_dim"], n_neurons=cfg["vocab_size"]),
        speaker_feature_extractor=Fbank(**fb),
        speaker_normalizer=InputNormalization(norm_type="sentence", update_until_epoch=4),
        speaker_frontend=fe("same"),
        speaker_encoder=enc(cfg["speaker_num_layers"]),
        speaker_proj=Linear(input_size=cfg["d_model"], n_neurons=cfg["d_model"]),
    )
    for name, mod in m.items():
        load_det_weights(mod, name + ".")
        mod.eval()
    return m


def forward_chain(m, inp, injection_mode, collect=None):
    """Restates what TSASR.compute_forward does with the reference's own modules
    (train_librispeechmix_scratch.py:34-148), stage by stage, collecting intermediates."""
    from speechbrain.dataio.dataio import length_to_mask

    c = {} if collect is None else collect
    mix, mix_l = T(inp["mixed_sig"]), T(inp["mixed_lens"])
    enr, enr_l = T(inp["enroll_sig"]), T(inp["enroll_lens"])
    tb, tb_l = T(inp["tokens_bos"]), T(inp["tokens_bos_lens"])

    sf = m["speaker_feature_extractor"](enr)
    c["spk_fbank"] = N(sf)
    sf = m["speaker_normalizer"](sf.clone(), enr_l, epoch=0)
    c["spk_norm"] = N(sf)
    sf = m["speaker_frontend"](sf)
    se = m["speaker_encoder"](sf, enr_l)
    c["spk_enc"] = N(se)
    if injection_mode != "cross_attention":
        mask = length_to_mask((enr_l * se.shape[-2]).ceil().clamp(max=se.shape[-2]).int())[..., None]
        se = se * mask
        se = se.sum(dim=-2, keepdims=True)
        se = se / mask.sum(dim=-2, keepdims=True)
    c["spk_pool"] = N(se)
    se = m["speaker_proj"](se)
    c["spk_emb"] = N(se)

    f = m["feature_extractor"](mix)
    c["fbank"] = N(f)
    f = m["normalizer"](f.clone(), mix_l, epoch=0)
    c["norm"] = N(f)
    f = m["frontend"](f)
    c["frontend"] = N(f)
    e = m["encoder"](f, mix_l, se, enr_l)
    c["enc"] = N(e)
    e = m["encoder_proj"](e)
    c["enc_proj"] = N(e)
    emb = m["embedding"](tb)
    d, _ = m["decoder"](emb, lengths=tb_l)
    c["dec"] = N(d)
    d = m["decoder_proj"](d)
    c["dec_proj"] = N(d)
    j = m["joiner"](e[..., None, :], d[:, None, ...])
    logits = m["transducer_head"](j)
    c["logits"] = N(logits)
    return logits, e, c


def main():
    torch.manual_seed(0)
    import_reference()
    from speechbrain.decoders.transducer import TransducerBeamSearcher
    from speechbrain.lobes.models.transformer.Conformer import ConformerEncoderLayer, ConvolutionModule
    from speechbrain.nnet.attention import RelPosEncXL, RelPosMHAXL
    from speechbrain.lobes.models.transformer.Transformer import get_lookahead_mask
    from speechbrain.dataio.dataio import length_to_mask

    os.makedirs(OUT, exist_ok=True)
    cfg = CFG1
    inp = golden_inputs(cfg)
    D, H = cfg["d_model"], cfg["nhead"]

    # ---------------- full chain, cat / non-causal: every stage ----------------
    with torch.no_grad():
        m = build(cfg, "cat", False, "same")
        logits, enc_out, c = forward_chain(m, inp, "cat")
        gs = TransducerBeamSearcher(
            decode_network_lst=[m["embedding"], m["decoder"], m["decoder_proj"]], tjoint=m["joiner"],
            classifier_network=[m["transducer_head"]], blank_id=0, beam_size=1, nbest=1)
        hyps, _, _, _ = gs(enc_out)
    hyp_len = np.array([len(h) for h in hyps], np.int64)
    hyp_pad = np.zeros((len(hyps), max(1, hyp_len.max())), np.int64)
    for i, h in enumerate(hyps):
        hyp_pad[i, : len(h)] = h
    np.savez_compressed(
        os.path.join(OUT, "c1_features.npz"),
        fbank=c["fbank"], norm=c["norm"], spk_fbank=c["spk_fbank"], spk_norm=c["spk_norm"])
    np.savez_compressed(
        os.path.join(OUT, "c1_chain_cat.npz"),
        frontend_b03=c["frontend"][[0, 3]], enc=c["enc"], enc_proj=c["enc_proj"], spk_enc=c["spk_enc"],
        spk_pool=c["spk_pool"], spk_emb=c["spk_emb"], dec=c["dec"], dec_proj=c["dec_proj"], logits=c["logits"],
        greedy_hyps=hyp_pad, greedy_lens=hyp_len)

    # ---------------- full chain backward with a probe (cat / non-causal) ----------------
    m = build(cfg, "cat", False, "same")
    logits, _, _ = forward_chain(m, inp, "cat")
    probe = T(det_tensor("probe.logits", logits.shape, 1.0))
    (logits * probe).sum().mul(1.0 / logits.numel()).backward()
    g = {}
    for mn, mod in m.items():
        for pn, p in mod.named_parameters():
            if p.grad is None:
                continue
            key = mn + "." + pn
            g["norm:" + key] = np.float64(p.grad.double().norm().item())
            if p.grad.numel() <= 4096:
                g["grad:" + key] = N(p.grad)
    np.savez_compressed(os.path.join(OUT, "c1_chain_cat_grads.npz"), **g)

    # ---------------- encoder variants: 4 injection modes x {non-causal, causal} ----------------
    ev = {}
    with torch.no_grad():
        for mode in ("cat", "sum", "prod", "cross_attention"):
            for causal in (False, True):
                mm = build(cfg, mode, causal, "causal" if causal else "same")
                lg, _, cc = forward_chain(mm, inp, mode)
                tag = f"{mode}{'_causal' if causal else ''}"
                ev["enc:" + tag] = cc["enc"]
                ev["logits_b1:" + tag] = cc["logits"][1]
                if mode == "sum" and causal:
                    ev["frontend_causal_b0"] = cc["frontend"][0]
    np.savez_compressed(os.path.join(OUT, "c1_encoder_variants.npz"), **ev)

    # ---------------- single blocks with gradients ----------------
    Tq = 50
    x_np = det_tensor("blk.x", (cfg["B"], Tq, D), 1.0)
    lens = T(np.asarray(cfg["mix_lens"], np.float32))
    kpm = ~length_to_mask((lens * Tq).round()).bool()
    blk = {}

    pe_mod = RelPosEncXL(D)
    with torch.no_grad():
        pe = pe_mod(T(x_np))
    blk["relpos_table"] = N(pe)

    def run(mod, fn, tag, probe_name):
        x = T(x_np).clone().requires_grad_(True)
        mod.zero_grad()
        y = fn(mod, x)
        pr = T(det_tensor(probe_name, y.shape, 1.0))
        (y * pr).sum().backward()
        blk[f"{tag}:out"] = N(y)
        blk[f"{tag}:dx"] = N(x.grad)
        for pn, p in mod.named_parameters():
            if p.grad is not None:
                blk[f"{tag}:d.{pn}"] = N(p.grad)

    mha = load_det_weights(RelPosMHAXL(embed_dim=D, num_heads=H, dropout=0.0, mask_pos_future=False), "blk.mha.").eval()
    mha_c = load_det_weights(RelPosMHAXL(embed_dim=D, num_heads=H, dropout=0.0, mask_pos_future=True), "blk.mha.").eval()
    run(mha, lambda md, x: md(x, x, x, pe)[0], "mha_nomask", "probe.blk")
    run(mha, lambda md, x: md(x, x, x, pe, key_padding_mask=kpm)[0], "mha_kpm", "probe.blk")
    cm = get_lookahead_mask(T(x_np))
    run(mha_c, lambda md, x: md(x, x, x, pe, key_padding_mask=kpm, attn_mask=cm)[0], "mha_kpm_causal", "probe.blk")
    with torch.no_grad():
        blk["mha_kpm:attn_b1"] = N(mha(T(x_np), T(x_np), T(x_np), pe, key_padding_mask=kpm)[1][1])

    conv = load_det_weights(ConvolutionModule(D, 31, True, torch.nn.LeakyReLU, 0.0, causal=False), "blk.conv.").eval()
    conv_c = load_det_weights(ConvolutionModule(D, 31, True, torch.nn.LeakyReLU, 0.0, causal=True), "blk.conv.").eval()
    run(conv, lambda md, x: md(x, kpm.unsqueeze(-1)), "conv", "probe.blk")
    run(conv_c, lambda md, x: md(x, kpm.unsqueeze(-1)), "conv_causal", "probe.blk")

    def mk_layer(causal):
        return load_det_weights(
            ConformerEncoderLayer(d_model=D, d_ffn=cfg["d_ffn"], nhead=H, kernel_size=31, activation=torch.nn.LeakyReLU,
                                  dropout=0.0, causal=causal), "blk.layer.").eval()

    lay, lay_c = mk_layer(False), mk_layer(True)
    run(lay, lambda md, x: md(x, src_key_padding_mask=kpm, pos_embs=pe)[0], "layer", "probe.blk")
    run(lay_c, lambda md, x: md(x, src_mask=cm, src_key_padding_mask=kpm, pos_embs=pe)[0], "layer_causal", "probe.blk")
    run(lay, lambda md, x: md.ffn_module1(x), "ffn", "probe.blk")
    # keep only small parameter grads for the layer (the big FFN ones are covered by "ffn")
    blk = {k: v for k, v in blk.items() if not (k.startswith("layer") and ":d." in k and v.size > 30000)}
    np.savez_compressed(os.path.join(OUT, "c1_blocks.npz"), x=x_np, **blk)

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("golden written to", OUT, "total bytes", tot)
    print("greedy hyps", hyps)


if __name__ == "__main__":
    main()
