#!/usr/bin/env python3
"""Golden vectors of the recipe WITHOUT a speaker encoder (TEST INFRASTRUCTURE; runs ONLY in the build container).

train_librispeechmix_none.py:34-95 with hparams/LibriSpeechMix/conformer-t_none.yaml: the speaker branch is gone and the encoder is built
WITHOUT the `injection_mode` / `injection_after` arguments (conformer-t_none.yaml:156-165 - the constructor defaults apply) and called as
`encoder(feats, mixed_sigs_lens)` (train_librispeechmix_none.py:78). The reference's own modules (stubs as in gen_golden.py), the
deterministic weights and inputs of golden_recipe.py, configs[0] sizes; outputs + gradient norms -> tests/golden/c1_none.npz.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_none.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import N, OUT, T, build, import_reference  # noqa: E402
from golden_recipe import CFG1, det_tensor, golden_inputs, load_det_weights  # noqa: E402


def main():
    torch.manual_seed(0)
    import_reference()
    from models.conformer import ConformerEncoder
    cfg = CFG1
    inp = golden_inputs(cfg)
    out = {}
    for causal in (False, True):
        m = build(cfg, "cat", causal, "same")
        for k in ("speaker_feature_extractor", "speaker_normalizer", "speaker_frontend", "speaker_encoder", "speaker_proj"):
            m.pop(k)
        enc = ConformerEncoder(input_size=cfg["encoder_input_size"], d_model=cfg["d_model"], nhead=cfg["nhead"],
                               num_layers=cfg["encoder_num_layers"], d_ffn=cfg["d_ffn"], dropout=0.0, activation=torch.nn.LeakyReLU,
                               kernel_size=cfg["kernel_size"], causal=causal)          # no injection arguments: conformer-t_none.yaml:156-165
        m["encoder"] = load_det_weights(enc, "encoder.").eval()
        mix, mix_l = T(inp["mixed_sig"]), T(inp["mixed_lens"])
        tb, tb_l = T(inp["tokens_bos"]), T(inp["tokens_bos_lens"])
        f = m["feature_extractor"](mix)
        f = m["normalizer"](f.clone(), mix_l, epoch=0)
        f = m["frontend"](f)
        e = m["encoder"](f, mix_l)
        tag = "causal" if causal else "full"
        out[f"enc:{tag}"] = N(e)
        e = m["encoder_proj"](e)
        d, _ = m["decoder"](m["embedding"](tb), lengths=tb_l)
        d = m["decoder_proj"](d)
        logits = m["transducer_head"](m["joiner"](e[..., None, :], d[:, None, ...]))
        out[f"logits:{tag}"] = N(logits)
        if not causal:
            probe = T(det_tensor("probe.logits", logits.shape, 1.0))
            (logits * probe).sum().mul(1.0 / logits.numel()).backward()
            for mn, mod in m.items():
                for pn, p in mod.named_parameters():
                    if p.grad is not None:
                        out[f"norm:{mn}.{pn}"] = np.float64(p.grad.double().norm().item())
    out["state_keys"] = np.asarray(sorted(f"{mn}.{k}" for mn, mod in m.items() for k in mod.state_dict()))
    np.savez_compressed(os.path.join(OUT, "c1_none.npz"), **out)
    print("written", os.path.join(OUT, "c1_none.npz"), os.path.getsize(os.path.join(OUT, "c1_none.npz")), "keys", len(out["state_keys"]))


if __name__ == "__main__":
    main()
