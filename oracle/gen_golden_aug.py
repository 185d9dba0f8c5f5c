#!/usr/bin/env python3
"""Golden vectors for the two augmenters inside TSASR.compute_forward (TEST INFRASTRUCTURE; runs ONLY in the build container).

train_librispeechmix_scratch.py:82-94 applies, in TRAIN stage with ``augment: True``,
  * ``speed_perturb``  = speechbrain.processing.speech_augmentation.SpeedPerturb(16000, speeds=[95,100,105]) to the mixture
    waveform (conformer-t_scratch.yaml:144-146), and
  * ``augmentation``   = speechbrain.lobes.augment.SpecAugment(time_warp window 5 bicubic, 2 freq masks < 30, 2 time masks < 20,
    fill with the mean) to the normalised features (conformer-t_scratch.yaml:132-142).
The reference is imported read-only from /root/reference (same stubs as gen_golden.py). Its random draws come from torch's
global generators; this script records them by wrapping ``torch.randint`` while the reference runs, so that the fixture holds
(input, draws, output) triples: the oracle restatement and the HIP kernels are checked with the SAME draws (their own
generators can never reproduce torch's streams, so parity of the draws themselves is statistical: ranges + distribution).

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_aug.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402


class RandintTap:
    """Records every torch.randint result drawn while active."""

    def __enter__(self):
        self.draws, self._orig = [], torch.randint

        def tapped(*a, **k):
            r = self._orig(*a, **k)
            self.draws.append(r.detach().cpu().clone())
            return r

        torch.randint = tapped
        return self

    def __exit__(self, *exc):
        torch.randint = self._orig


def main():
    G.import_reference()
    from speechbrain.lobes.augment import SpecAugment
    from speechbrain.processing.speech_augmentation import Resample, SpeedPerturb

    out = {}
    g = torch.Generator().manual_seed(77)
    # ---- SpecAugment: recipe settings on configs[0]-sized features [4,200,80], plus replace_with_zero and a too-short input ----
    x = torch.randn(4, 200, 80, generator=g)
    cases = {
        "recipe": dict(time_warp=True, time_warp_window=5, time_warp_mode="bicubic", freq_mask=True, n_freq_mask=2, time_mask=True,
                       n_time_mask=2, replace_with_zero=False, freq_mask_width=30, time_mask_width=20),
        "zero": dict(time_warp=True, time_warp_window=5, freq_mask=True, n_freq_mask=2, time_mask=True, n_time_mask=2,
                     replace_with_zero=True, freq_mask_width=(0, 20), time_mask_width=(0, 100)),
        "nowarp": dict(time_warp=False, freq_mask=True, n_freq_mask=3, time_mask=True, n_time_mask=1, replace_with_zero=False,
                       freq_mask_width=(5, 12), time_mask_width=(1, 40)),
    }
    for name, kw in cases.items():
        for rep in range(3):
            torch.manual_seed(100 + rep)
            aug = SpecAugment(**kw)
            with RandintTap() as tap:
                y = aug(x.clone())
            d = [t.reshape(-1).numpy().astype(np.int64) for t in tap.draws]
            key = f"sa_{name}_{rep}"
            if kw.get("time_warp", True):
                out[key + "_c"], out[key + "_w"] = d[0], d[1] + 1  # the reference adds 1 to its second draw (augment.py:136)
                d = d[2:]
            out[key + "_flen"], out[key + "_fpos"], out[key + "_tlen"], out[key + "_tpos"] = d[0], d[1], d[2], d[3]
            out[key + "_y"] = y.numpy()
    out["sa_x"] = x.numpy()
    # time - window <= window: the warp is skipped (augment.py:131-132)
    xs = torch.randn(2, 10, 80, generator=g)
    torch.manual_seed(5)
    aug = SpecAugment(time_warp=True, time_warp_window=5, freq_mask=False, time_mask=False)
    out["sa_short_x"], out["sa_short_y"] = xs.numpy(), aug(xs.clone()).numpy()

    # ---- SpeedPerturb / Resample: the recipe's three rates + the reference's own half-speed unit test shape ----
    wav = torch.randn(3, 4000, generator=g) * 0.1
    out["sp_x"] = wav.numpy()
    for speed in (95, 100, 105, 50):
        rs = Resample(orig_freq=16000, new_freq=16000 * speed // 100)
        out[f"sp_{speed}_y"] = rs(wav.clone()).numpy()
        if speed != 100:
            out[f"sp_{speed}_first"], out[f"sp_{speed}_weights"] = rs.first_indices.numpy(), rs.weights.numpy()
    torch.manual_seed(11)
    sp = SpeedPerturb(16000, speeds=[95, 100, 105])
    idx = []
    for _ in range(64):
        with RandintTap() as tap:
            sp(wav[:, :400])
        idx.append(int(tap.draws[0]))
    out["sp_index_draws"] = np.array(idx, np.int64)
    sine = torch.sin(torch.arange(16000.0)).unsqueeze(0)       # tests/unittests/test_augment.py:100-113
    out["sp_sine_half"] = Resample(16000, 8000)(sine).numpy()
    np.savez_compressed(os.path.join(G.OUT, "c1_augment.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
