#!/usr/bin/env python3
"""Golden batch compositions of the reference's DynamicBatchSampler (TEST INFRASTRUCTURE; runs ONLY in the build container).

speechbrain/dataio/sampler.py:306-702 imported read-only from /root/reference (stubs as in gen_golden.py), fed with a seeded list
of utterance durations shaped like LibriSpeechMix (2-35 s) and the recipe's arguments (train_librispeechmix_scratch.py:577-600,
conformer-t_scratch.yaml:64-72: max_batch_length 50 s, num_buckets 80, shuffle False, ordering ascending / descending, plus the
shuffled / random-order / max_batch_ex / explicit-boundary variants). Stores every batch as a flat index list + offsets in
tests/golden/c1_sampler.npz.

Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_sampler.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

CASES = {
    "train": dict(max_batch_length=50.0, num_buckets=80, shuffle=False, batch_ordering="ascending", max_batch_ex=None),
    "valid": dict(max_batch_length=50.0, num_buckets=80, shuffle=False, batch_ordering="descending", max_batch_ex=6),
    "shuffled_e0": dict(max_batch_length=30.0, num_buckets=12, shuffle=True, batch_ordering="random", seed=7, epoch=0),
    "shuffled_e3": dict(max_batch_length=30.0, num_buckets=12, shuffle=True, batch_ordering="random", seed=7, epoch=3),
    "boundaries": dict(max_batch_length=40.0, bucket_boundaries=[5.0, 10.0, 20.0, 40.0], shuffle=False, batch_ordering="descending",
                       drop_last=True),
}


def durations(n=500, seed=123):
    rng = np.random.RandomState(seed)
    return np.round(np.clip(rng.lognormal(2.5, 0.5, n), 2.0, 35.0), 2)


def main():
    G.import_reference()
    from speechbrain.dataio.sampler import DynamicBatchSampler

    lens = durations()
    out = {"durations": lens}
    for name, kw in CASES.items():
        s = DynamicBatchSampler(list(range(len(lens))), lengths_list=lens.tolist(), **kw)
        batches = [list(b) for b in s]
        out[name + "_flat"] = np.array([i for b in batches for i in b], np.int64)
        out[name + "_sizes"] = np.array([len(b) for b in batches], np.int64)
        out[name + "_boundaries"] = np.asarray(s._bucket_boundaries, np.float64)
        print(name, len(batches), "batches, sizes", sorted(set(len(b) for b in batches))[:8])
    np.savez_compressed(os.path.join(G.OUT, "c1_sampler.npz"), **out)


if __name__ == "__main__":
    main()
