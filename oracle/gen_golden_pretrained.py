#!/usr/bin/env python3
"""Golden vectors of the pretrained-speaker variant (TEST INFRASTRUCTURE; runs ONLY in the build container).

BASELINE.json configs[3]: train_librispeechmix_pretrained.py with hparams/LibriSpeechMix/conformer-t_wavlm.yaml. The frozen WavLM
x-vector model itself is out of scope (needs a download): what is on the hot path is everything behind its
output - ``speaker_proj`` Linear(512 -> d_model) (conformer-t_wavlm.yaml:203-206), the injection into the encoder and the transducer.
This script imports the reference's own modules (same stubs as gen_golden.py), builds them with the wavlm YAML's constructor
arguments at the configs[0] sizes, feeds the deterministic embedding of ``golden_recipe.golden_enroll_emb`` where the reference feeds
``speaker_encoder(...).embeddings[:, None, :]`` (train_librispeechmix_pretrained.py:59-63,80) and dumps outputs + gradient norms to
tests/golden/c1_pretrained.npz.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_pretrained.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import N, OUT, T, build, import_reference  # noqa: E402
from golden_recipe import CFG1, SPEAKER_EMBEDDING_DIM, det_tensor, golden_enroll_emb, golden_inputs, load_det_weights  # noqa: E402


def forward_chain(m, inp, emb):
    """train_librispeechmix_pretrained.py:36-135 with the reference's own modules, speaker_encoder output given."""
    mix, mix_l, enr_l = T(inp["mixed_sig"]), T(inp["mixed_lens"]), T(inp["enroll_lens"])
    tb, tb_l = T(inp["tokens_bos"]), T(inp["tokens_bos_lens"])
    c = {}
    se = m["speaker_proj"](T(emb))
    c["spk_emb"] = N(se)
    f = m["feature_extractor"](mix)
    f = m["normalizer"](f.clone(), mix_l, epoch=0)
    f = m["frontend"](f)
    e = m["encoder"](f, mix_l, se, enr_l)
    c["enc"] = N(e)
    e = m["encoder_proj"](e)
    d, _ = m["decoder"](m["embedding"](tb), lengths=tb_l)
    d = m["decoder_proj"](d)
    logits = m["transducer_head"](m["joiner"](e[..., None, :], d[:, None, ...]))
    c["logits"] = N(logits)
    return logits, c


def main():
    torch.manual_seed(0)
    import_reference()
    from speechbrain.nnet.linear import Linear
    cfg = CFG1
    inp, emb = golden_inputs(cfg), golden_enroll_emb(cfg)
    out = {}
    for mode in ("cat", "sum", "prod"):
        m = build(cfg, mode, False, "same")
        for k in ("speaker_feature_extractor", "speaker_normalizer", "speaker_frontend", "speaker_encoder"):
            m.pop(k)                                    # conformer-t_wavlm.yaml has none of them (:199-229 of the scratch YAML removed)
        m["speaker_proj"] = load_det_weights(Linear(input_size=SPEAKER_EMBEDDING_DIM, n_neurons=cfg["d_model"]), "speaker_proj.").eval()
        logits, c = forward_chain(m, inp, emb)
        out[f"spk_emb:{mode}"], out[f"enc:{mode}"], out[f"logits:{mode}"] = c["spk_emb"], c["enc"], c["logits"]
        if mode == "cat":   # gradients of every trainable parameter for a fixed probe
            probe = T(det_tensor("probe.logits", logits.shape, 1.0))
            (logits * probe).sum().mul(1.0 / logits.numel()).backward()
            for mn, mod in m.items():
                for pn, p in mod.named_parameters():
                    if p.grad is not None:
                        out[f"norm:{mn}.{pn}"] = np.float64(p.grad.double().norm().item())
                        if p.grad.numel() <= 4096:
                            out[f"grad:{mn}.{pn}"] = N(p.grad)
    np.savez_compressed(os.path.join(OUT, "c1_pretrained.npz"), **out)
    print("written", os.path.join(OUT, "c1_pretrained.npz"), os.path.getsize(os.path.join(OUT, "c1_pretrained.npz")))


if __name__ == "__main__":
    main()
