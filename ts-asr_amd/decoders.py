"""Transducer search with the reference's constructor/return shape (speechbrain/decoders/transducer.py:14-373).

beam_size = 1 -> greedy (transducer.py:138-218): at most one symbol per encoder frame; the predictor state of an utterance
advances only when it emitted a non-blank. The per-frame decision stays on the device (argmax + masked state update, no
per-item Python loop as in transducer.py:187-194); only the final token table is read back.
beam_size > 1 -> the reference's per-utterance beam search with state_beam / expand_beam pruning (transducer.py:220-373,
no LM fusion): host-side hypothesis bookkeeping exactly as specified there, device-side predictor /
joint / head steps. Pinned to the reference's own hypotheses by tests/golden/c1_beam.npz.
"""
import os

import torch
import torch.nn.functional as F


class TransducerBeamSearcher(torch.nn.Module):
    def __init__(self, decode_network_lst, tjoint, classifier_network, blank_id, beam_size=4, nbest=5, lm_module=None,
                 lm_weight=0.0, state_beam=2.3, expand_beam=2.3):
        super().__init__()
        self.decode_network_lst, self.tjoint, self.classifier_network = decode_network_lst, tjoint, classifier_network
        self.blank_id, self.beam_size, self.nbest = blank_id, beam_size, nbest
        self.state_beam, self.expand_beam = state_beam, expand_beam
        if lm_module is not None or lm_weight != 0.0:
            raise NotImplementedError("LM fusion is not part of the TS-ASR recipes")

    def forward(self, tn_output):
        if self.beam_size <= 1:
            return self.transducer_greedy_decode(tn_output)
        return self.transducer_beam_search_decode(tn_output)

    def _pn(self, tok, hidden):
        emb, dec, proj = self.decode_network_lst
        out, hidden = dec(emb(tok), hx=hidden)
        return proj(out), hidden

    def _device_greedy_ok(self, tn_output):
        """The one-launch device decoder (csrc/search.hip) covers the recipes' networks: one-hot or learned embedding (<= 64 columns),
        one-layer unidirectional LSTM, Linear projection, joint = LeakyReLU(sum), one Linear classifier."""
        from . import nnet, rnnt
        if os.environ.get("TSASR_GREEDY_KERNEL", "1") == "0" or not tn_output.is_cuda or tn_output.dtype not in (torch.float32, torch.bfloat16):
            return False
        if len(self.decode_network_lst) != 3 or len(self.classifier_network) != 1:
            return False
        emb, dec, proj = self.decode_network_lst
        head = self.classifier_network[0]
        ok = (isinstance(emb, nnet.Embedding) and isinstance(dec, nnet.LSTM) and isinstance(proj, nnet.Linear) and isinstance(head, nnet.Linear)
              and isinstance(self.tjoint, rnnt.Transducer_joint) and isinstance(self.tjoint.nonlinearity, torch.nn.LeakyReLU)
              and dec.rnn.num_layers == 1 and not dec.rnn.bidirectional and emb.embedding_dim <= 64 and head.w.out_features <= 63
              and dec.rnn.hidden_size % 4 == 0 and proj.w.out_features % 4 == 0 and proj.w.out_features == tn_output.shape[-1])
        return bool(ok)

    @torch.no_grad()
    def _greedy_on_device(self, tn_output):
        from . import _capi as C
        emb, dec, proj = self.decode_network_lst
        head = self.classifier_network[0]
        B, T, J = tn_output.shape
        enc = tn_output.contiguous()
        f = lambda t: None if t is None else t.detach().float().contiguous()  # noqa: E731
        rnn = dec.rnn
        mats = [rnn.weight_ih_l0, rnn.weight_hh_l0, proj.w.weight, head.w.weight]
        if enc.dtype == torch.bfloat16:       # the training step's bf16 shadows (or a cast) - half the bytes per predictor step
            from .ops import _bf16_weight
            mats, wdt = [_bf16_weight(m).contiguous() for m in mats], C.BF16
        else:
            mats, wdt = [f(m) for m in mats], C.F32
        table = f(emb.Embedding.weight)
        b_ih, b_hh = (f(rnn.bias_ih_l0), f(rnn.bias_hh_l0)) if rnn.bias else (None, None)
        b_proj, b_head = f(proj.w.bias), f(head.w.bias)
        preds = torch.empty(B, T, dtype=torch.int32, device=enc.device)
        logp = torch.empty(B, dtype=torch.float32, device=enc.device)
        C.check(C.lib().tsasr_greedy_decode(C.ptr(enc), C.ptr(table), C.ptr(mats[0]), C.ptr(mats[1]), C.ptr(b_ih), C.ptr(b_hh), C.ptr(mats[2]),
                                            C.ptr(b_proj), C.ptr(mats[3]), C.ptr(b_head), C.ptr(preds), C.ptr(logp), B, T, J, rnn.hidden_size,
                                            table.shape[1], mats[3].shape[0], int(self.blank_id), float(self.tjoint.nonlinearity.negative_slope),
                                            C.io_dtype(enc), wdt, C.stream_ptr()), "tsasr_greedy_decode")
        rows = preds.cpu()
        hyps = [[int(v) for v in row[row >= 0]] for row in rows]
        return hyps, logp.exp().mean(), None, None

    @torch.no_grad()
    def transducer_greedy_decode(self, tn_output):
        if self._device_greedy_ok(tn_output):
            return self._greedy_on_device(tn_output)
        B, T, _ = tn_output.shape
        dev = tn_output.device
        tok = torch.full((B, 1), self.blank_id, dtype=torch.long, device=dev)
        out_pn, hidden = self._pn(tok, None)
        preds = torch.full((B, T), -1, dtype=torch.long, device=dev)
        logp_sum = torch.zeros(B, device=dev)
        for t in range(T):
            j = self.tjoint(tn_output[:, t, :].unsqueeze(1).unsqueeze(1), out_pn.unsqueeze(1))
            for layer in self.classifier_network:
                j = layer(j)
            logp, pos = torch.max(F.log_softmax(j.float(), dim=-1).squeeze(1).squeeze(1), dim=1)
            upd = pos != self.blank_id
            preds[:, t] = torch.where(upd, pos, preds[:, t])
            logp_sum = logp_sum + torch.where(upd, logp, torch.zeros_like(logp))
            new_tok = torch.where(upd, pos, tok[:, 0]).unsqueeze(1)
            new_out, new_hidden = self._pn(new_tok, hidden)
            m = upd.view(B, 1, 1)
            out_pn = torch.where(m, new_out, out_pn)
            hidden = tuple(torch.where(upd.view(1, B, 1), nh, h) for nh, h in zip(new_hidden, hidden))
            tok = new_tok
        table = preds.cpu()
        hyps = [[int(x) for x in row[row >= 0]] for row in table]
        return hyps, logp_sum.exp().mean(), None, None

    @torch.no_grad()
    def transducer_beam_search_decode(self, tn_output):
        """Returns (best hyps, mean exp(normalised score), n-best hyps, n-best normalised log-scores) like the reference.
        A = hypotheses still to be extended at this frame, B = those that emitted blank here (the next frame's A). Until
        |B| >= beam: take the best a in A by logp / len(prediction); stop once the best b in B has logp >= state_beam + logp(a);
        run the predictor on a's last token, score the beam best symbols of the joint at this frame; blank closes a copy of
        a into B, a non-blank symbol within expand_beam of the best non-blank extends a (new predictor state) back into A."""
        dev = tn_output.device
        key = lambda hyp: hyp[1] / len(hyp[0])  # noqa: E731
        nbest_batch, nbest_scores = [], []
        for b in range(tn_output.shape[0]):
            beam = [([self.blank_id], 0.0, None)]           # (prediction incl. the blank prefix, logp, predictor state)
            for t in range(tn_output.shape[1]):
                A, beam = beam, []
                frame = tn_output[b, t, :].view(1, 1, 1, -1)
                while len(beam) < self.beam_size:
                    a = max(A, key=key)
                    if beam and max(beam, key=key)[1] >= self.state_beam + a[1]:
                        break
                    A.remove(a)
                    tok = torch.full((1, 1), a[0][-1], dtype=torch.long, device=dev)
                    out_pn, new_state = self._pn(tok, a[2])
                    j = self.tjoint(frame, out_pn.unsqueeze(0))
                    for layer in self.classifier_network:
                        j = layer(j)
                    logp, pos = torch.topk(F.log_softmax(j.float(), dim=-1).view(-1), k=self.beam_size)
                    logp, pos = logp.tolist(), pos.tolist()      # one host read per expansion (the reference: one per symbol)
                    best_nonblank = logp[0] if pos[0] != self.blank_id else logp[1]
                    for lp, sym in zip(logp, pos):
                        if sym == self.blank_id:
                            beam.append((a[0][:], a[1] + lp, a[2]))
                        elif lp >= best_nonblank - self.expand_beam:
                            A.append((a[0] + [sym], a[1] + lp, new_state))
            ranked = sorted(beam, key=key, reverse=True)[: self.nbest]
            nbest_batch.append([h[0][1:] for h in ranked])
            nbest_scores.append([h[1] / len(h[0]) for h in ranked])
        best = [n[0] for n in nbest_batch]
        return best, torch.tensor([s_[0] for s_ in nbest_scores]).exp().mean(), nbest_batch, nbest_scores
