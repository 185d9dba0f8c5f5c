#!/usr/bin/env python3
"""Golden vectors of the FULL-WIDTH model (oracle/golden_recipe.CFG2: d_model 256, Dh 64, d_ffn 2048, joint 640, predictor 512; 2 + 2
layers, B = 2) from the reference itself (TEST INFRASTRUCTURE; runs ONLY in the build container, as oracle/gen_golden.py does):
stage outputs of the `cat` / non-causal chain, greedy hypotheses, and - for a fixed linear probe on the logits, the RNN-T loss needs
torchaudio - the gradient NORM of every parameter plus the small gradients themselves.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_d256.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
from golden_recipe import CFG2, det_tensor, golden_inputs  # noqa: E402


def lattice_mask(cfg, shape):
    """1 inside each utterance's RNN-T lattice (t < round(len * T'), u <= round(len * U): SB/nnet/losses.py:58-59), 0 outside: the probe looks
    only at cells the training path computes (the build's fused joint leaves the cells outside the lattice unwritten)."""
    B, Tp, U1, _ = shape
    m = torch.zeros(B, Tp, U1, 1)
    for b in range(B):
        tb = int(round(float(cfg["mix_lens"][b]) * Tp))
        ub = int(round(float(cfg["tok_lens"][b]) * (U1 - 1)))
        m[b, :tb, : ub + 1] = 1.0
    return m


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    G.import_reference()
    from speechbrain.decoders.transducer import TransducerBeamSearcher
    cfg = CFG2
    inp = golden_inputs(cfg)
    with torch.no_grad():
        m = G.build(cfg, "cat", False, "same")
        logits, enc_out, c = G.forward_chain(m, inp, "cat")
        gs = TransducerBeamSearcher(decode_network_lst=[m["embedding"], m["decoder"], m["decoder_proj"]], tjoint=m["joiner"],
                                    classifier_network=[m["transducer_head"]], blank_id=0, beam_size=1, nbest=1)
        hyps, _, _, _ = gs(enc_out)
    hyp_len = np.array([len(h) for h in hyps], np.int64)
    hyp_pad = np.zeros((len(hyps), max(1, hyp_len.max())), np.int64)
    for i, h in enumerate(hyps):
        hyp_pad[i, : len(h)] = h
    out = dict(norm=c["norm"], spk_norm=c["spk_norm"], spk_emb=c["spk_emb"], enc=c["enc"], enc_proj=c["enc_proj"], dec_proj=c["dec_proj"],
               logits=c["logits"], greedy_hyps=hyp_pad, greedy_lens=hyp_len)
    # backward of a fixed linear probe on the logits
    m = G.build(cfg, "cat", False, "same")
    logits, _, _ = G.forward_chain(m, inp, "cat")
    probe = G.T(det_tensor("probe.logits.c2", logits.shape, 1.0)) * lattice_mask(cfg, logits.shape)
    (logits * probe).sum().mul(1.0 / logits.numel()).backward()
    for mn, mod in m.items():
        for pn, p in mod.named_parameters():
            if p.grad is None:
                continue
            key = mn + "." + pn
            out["norm:" + key] = np.float64(p.grad.double().norm().item())
            if p.grad.numel() <= 2048:
                out["grad:" + key] = G.N(p.grad)
    np.savez_compressed(os.path.join(G.OUT, "c2_fullwidth.npz"), **out)
    print("c2_fullwidth.npz:", len(out), "arrays,", os.path.getsize(os.path.join(G.OUT, "c2_fullwidth.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
