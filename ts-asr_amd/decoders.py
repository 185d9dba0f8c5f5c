"""Greedy transducer search with the reference's constructor/return shape
(speechbrain/decoders/transducer.py:14-218; beam_size=1 => transducer_greedy_decode).

At most one symbol per encoder frame; the predictor state of an utterance advances only when it emitted a
non-blank. The per-frame decision stays on the device (argmax + masked state update, no per-item Python loop as in
transducer.py:187-194); only the final token table is read back. Beam search (beam_size > 1, transducer.py:220-373)
is SURVEY.md section 8f row f1 = next round.
"""
import torch
import torch.nn.functional as F


class TransducerBeamSearcher(torch.nn.Module):
    def __init__(self, decode_network_lst, tjoint, classifier_network, blank_id, beam_size=4, nbest=5, lm_module=None,
                 lm_weight=0.0, state_beam=2.3, expand_beam=2.3):
        super().__init__()
        self.decode_network_lst, self.tjoint, self.classifier_network = decode_network_lst, tjoint, classifier_network
        self.blank_id, self.beam_size, self.nbest = blank_id, beam_size, nbest
        if lm_module is not None or lm_weight != 0.0:
            raise NotImplementedError("LM fusion is not part of the TS-ASR recipes")

    def forward(self, tn_output):
        if self.beam_size <= 1:
            return self.transducer_greedy_decode(tn_output)
        raise NotImplementedError("beam search (beam_size > 1) is SURVEY.md section 8f row f1; use beam_size=1")

    def _pn(self, tok, hidden):
        emb, dec, proj = self.decode_network_lst
        out, hidden = dec(emb(tok), hx=hidden)
        return proj(out), hidden

    @torch.no_grad()
    def transducer_greedy_decode(self, tn_output):
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
