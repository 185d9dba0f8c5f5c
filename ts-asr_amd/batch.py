"""Batch object with the reference's surface (speechbrain/dataio/batch.py:17-190): tensor fields are
``PaddedData(data, lengths)`` 2-tuples with RELATIVE lengths, plus ``id`` / ``target_words`` lists and ``.to()``.
``synthetic_batch`` builds the LibriSpeechMix-shaped inputs the bench and the tests use (no dataset on the box).
"""
import collections

import torch

PaddedData = collections.namedtuple("PaddedData", ["data", "lengths"])


class PaddedBatch:
    def __init__(self, fields):
        self._keys = list(fields)
        for k, v in fields.items():
            setattr(self, k, v)

    def to(self, *args, **kwargs):
        for k in self._keys:
            v = getattr(self, k)
            if isinstance(v, PaddedData):
                setattr(self, k, PaddedData(v.data.to(*args, **kwargs), v.lengths.to(*args, **kwargs)))
            elif torch.is_tensor(v):
                setattr(self, k, v.to(*args, **kwargs))
        return self

    def pin_memory(self):
        for k in self._keys:
            v = getattr(self, k)
            if isinstance(v, PaddedData):
                setattr(self, k, PaddedData(v.data.pin_memory(), v.lengths.pin_memory()))
        return self

    def __iter__(self):
        return iter(getattr(self, k) for k in self._keys)


def synthetic_batch(B, n_mix, n_enroll, U, vocab_size=29, seed=1234, ragged=False, feats=False, n_mels=80, device="cpu", enroll_emb_dim=0):
    """Seeded synthetic TS-ASR batch. ``feats=False``: waveforms N(0, 0.1^2) of n_mix / n_enroll samples;
    ``feats=True``: already-normalised mel features N(0,1) of n_mix / n_enroll FRAMES
    carried in the same fields (the recipe skips Fbank+norm when hparams['input_is_feats'])."""
    g = torch.Generator().manual_seed(seed)
    if feats:
        mix = torch.randn(B, n_mix, n_mels, generator=g)
        enr = torch.randn(B, n_enroll, n_mels, generator=g)
    else:
        mix = torch.randn(B, n_mix, generator=g) * 0.1
        enr = torch.randn(B, n_enroll, generator=g) * 0.1
    tokens = torch.randint(1, vocab_size, (B, U), generator=g)
    if ragged:  # sorted ascending like the reference's `sorting: ascending`
        mix_l = torch.sort(torch.rand(B, generator=g) * 0.4 + 0.6).values
        mix_l[-1] = 1.0
        tok_l = mix_l.clone()
        enr_l = torch.rand(B, generator=g) * 0.5 + 0.5
        enr_l[0] = 1.0
    else:
        mix_l, tok_l, enr_l = torch.ones(B), torch.ones(B), torch.ones(B)
    tok_abs = (tok_l * U).round().long()
    for b in range(B):
        tokens[b, tok_abs[b]:] = 0
        mix[b, int(round(float(mix_l[b]) * n_mix)):] = 0
        enr[b, int(round(float(enr_l[b]) * n_enroll)):] = 0
    tokens_bos = torch.cat([torch.zeros(B, 1, dtype=torch.long), tokens], 1)
    bos_l = (tok_abs + 1).float() / (U + 1)
    fields = {
        "id": [f"syn-{seed}-{i}" for i in range(B)],
        "mixed_sig": PaddedData(mix, mix_l), "enroll_sig": PaddedData(enr, enr_l),
        "tokens_bos": PaddedData(tokens_bos, bos_l), "tokens": PaddedData(tokens, tok_l),
        "target_words": [["x"] for _ in range(B)],
    }
    if enroll_emb_dim:   # pretrained-speaker variant: the frozen encoder's x-vector N(0,1) [B,1,E] is an input
        fields["enroll_emb"] = PaddedData(torch.randn(B, 1, enroll_emb_dim, generator=g), torch.ones(B))
    batch = PaddedBatch(fields)
    return batch.to(device) if device != "cpu" else batch
