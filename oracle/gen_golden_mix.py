#!/usr/bin/env python3
"""Golden mixtures of the reference's audio pipeline (TEST INFRASTRUCTURE; runs ONLY in the build container).

train_librispeechmix_scratch.py:196-487 (`dataio_prepare`) is imported read-only from /root/reference (stub modules as in gen_golden.py) and
RUN: its `audio_pipeline` closure (:333-456 - gain of the non-target sources, delay padding, left-to-right sum, crop, enrollment trim)
is the reference's own code, reached through speechbrain's DynamicItemDataset exactly as the recipe reaches it. Only the two torchaudio
calls of the closure are stubbed, with the minimum they need to be: `torchaudio.load(path)` returns a seeded synthetic waveform per path
(there is no audio on this box) and `torchaudio.functional.resample(sig, sr, sr)` is the identity it is for equal rates.
Stores the source waveforms, the manifest fields and the reference's `mixed_sig` / `enroll_sig` per case in tests/golden/c1_mix.npz.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_mix.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

SR = 16000
# (name, gain_nontarget dB, trim_enroll s, entries); an entry: source lengths (samples), delays (s), start, duration, target index
CASES = [
    ("two_spk_gain0", 0, 20.0, [dict(lens=[8000, 10400], delays=[0.0, 0.2503], start=0.0, duration=0.80315, target=0),
                                dict(lens=[6011, 5600], delays=[0.173, 0.0], start=0.1001, duration=0.35, target=1)]),
    ("two_spk_gain_m5", -5, 0.3, [dict(lens=[8000, 10400], delays=[0.0, 0.2503], start=0.0, duration=0.80315, target=0),
                                  dict(lens=[6011, 5600], delays=[0.173, 0.0], start=0.1001, duration=0.35, target=1)]),
    ("three_spk_gain_p3", 3, 0.4001, [dict(lens=[4800, 6200, 5500], delays=[0.05, 0.0, 0.12001], start=0.02, duration=0.44, target=2),
                                      dict(lens=[3200, 3200, 3200], delays=[0.0, 0.0, 0.0], start=0.0, duration=0.2, target=0)]),
]


def wave(path):
    """Seeded synthetic source per file name: N(0, a^2) with a per-file amplitude (so the gain rule has something to do)."""
    seed = sum((i + 1) * ord(c) for i, c in enumerate(path)) % (2 ** 31)
    rng = np.random.RandomState(seed)
    n = int(path.rsplit("_n", 1)[1].split(".")[0])
    return (rng.standard_normal(n) * (0.02 + 0.1 * rng.rand())).astype(np.float32)


def main():
    G.import_reference()
    import torchaudio
    loaded = {}

    def load(path):
        loaded[path] = wave(path)
        return torch.from_numpy(loaded[path].copy())[None, :], SR

    def resample(sig, sr_in, sr_out):
        assert sr_in == sr_out == SR
        return sig

    torchaudio.load = load
    torchaudio.functional.resample = resample
    import train_librispeechmix_scratch as ref

    class Tok:
        class sp:
            @staticmethod
            def encode_as_ids(wrd):
                return [1 + (ord(c) % 27) for c in wrd if c != " "]

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, gain, trim, entries in CASES:
            man = {}
            for k, e in enumerate(entries):
                uid = f"{name}/utt{k}"
                man[uid] = {
                    "wavs": [f"{{DATA_ROOT}}/{name}_{k}_src{j}_n{n}.wav" for j, n in enumerate(e["lens"])],
                    "enroll_wav": f"{{DATA_ROOT}}/{name}_{k}_enroll_n{9000 + 500 * k}.wav",
                    "delays": e["delays"], "start": e["start"], "duration": e["duration"], "target_speaker_idx": e["target"],
                    "wrd": "hello world",
                }
            path = os.path.join(tmp, name + ".json")
            with open(path, "w") as f:
                json.dump(man, f)
            hp = dict(data_folder="/data", train_json=path, valid_json=path, test_json=path, sorting="random",
                      train_remove_if_longer=100.0, valid_remove_if_longer=100.0, test_remove_if_longer=100.0, sample_rate=SR,
                      gain_nontarget=gain, trim_enroll=trim, plot_data=False, prompt_test=False, prompt_mode=[], blank_index=0)
            train, _, _ = ref.dataio_prepare(hp, Tok())
            ids = list(train.data_ids)
            for k, e in enumerate(entries):
                uid = f"{name}/utt{k}"
                item = train[ids.index(uid)]
                pre = f"{name}.{k}."
                for j, p in enumerate(man[uid]["wavs"]):
                    out[pre + f"src{j}"] = loaded[p.replace("{DATA_ROOT}", "/data")]
                out[pre + "enroll_src"] = loaded[man[uid]["enroll_wav"].replace("{DATA_ROOT}", "/data")]
                out[pre + "delays"] = np.asarray(e["delays"], np.float64)
                out[pre + "meta"] = np.asarray([e["start"], e["duration"], e["target"], gain, trim, len(e["lens"])], np.float64)
                out[pre + "mixed_sig"] = item["mixed_sig"].numpy().copy()
                out[pre + "enroll_sig"] = item["enroll_sig"].numpy().copy()
                out[pre + "tokens_bos"] = item["tokens_bos"].numpy().copy()
                print(pre, "mixed", item["mixed_sig"].shape, "enroll", item["enroll_sig"].shape, "rms", float(item["mixed_sig"].pow(2).mean().sqrt()))
    out["cases"] = np.asarray([f"{name}.{k}" for name, _, _, entries in CASES for k in range(len(entries))])
    np.savez_compressed(os.path.join(G.OUT, "c1_mix.npz"), **out)


if __name__ == "__main__":
    main()
