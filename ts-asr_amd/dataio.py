"""Data side of the recipe: what turns a LibriSpeechMix manifest into the batches ``TSASR.fit_batch`` eats.

Host logic only - nothing here is on the device hot path; a maintainer may equally keep the reference's own ``speechbrain.dataio``
(pure CPU code) in front of the MI355X step. Mirrors:

  * ``DynamicBatchSampler``  - speechbrain/dataio/sampler.py:306-702 (length-bucketed batches: log-normal bucket boundaries,
    ``max_batch_length`` / boundary examples per bucket, batches emitted as buckets fill, then ordered). Batch compositions are
    identical to the reference's for the same arguments (tests/golden/c1_sampler.npz).
  * ``mix_sources`` / ``trim_enroll`` - the arithmetic of ``audio_pipeline`` (train_librispeechmix_scratch.py:338-386): per-speaker
    gain against the target's power, delay padding, sum, crop. Works on CPU or GPU tensors. The reference's version is a closure
    around torchaudio file reads, so it cannot be run here: this restatement is unpinned (plain tensor arithmetic, unit-tested
    against hand-computed cases only).
  * ``load_manifest`` - the JSON written by librispeechmix_prepare.py:206-218 ({id: {wavs, enroll_wav, delays, start, duration,
    durations, target_speaker_idx, wrd, speakers, genders}}) with the ``{data_folder}`` replacement of DynamicItemDataset.from_json.
  * ``collate`` - PaddedBatch of (mixed_sig, enroll_sig, tokens_bos, tokens) with relative lengths (speechbrain/dataio/batch.py:20-190).
Audio decoding (FLAC) and the SentencePiece tokenizer stay with the caller.
"""
import json
import math

import numpy as np
import torch
from torch.utils.data import Sampler

from .batch import PaddedBatch, PaddedData


class DynamicBatchSampler(Sampler):
    def __init__(self, dataset, max_batch_length, num_buckets=None, length_func=lambda x: x["duration"], shuffle=True,
                 batch_ordering="random", max_batch_ex=None, bucket_boundaries=(), lengths_list=None, seed=42, epoch=0,
                 drop_last=False, verbose=False):
        self._dataset = dataset
        if num_buckets is None and len(bucket_boundaries) == 0:
            raise RuntimeError("Please specify either num_buckets or bucket boundaries.Check the docs, and/or the tutorial !")
        if lengths_list is not None:
            self._ex_lengths = [lengths_list[i] for i in range(len(lengths_list))]
        else:
            if not (hasattr(dataset, "data") and hasattr(dataset, "data_ids")):
                raise NotImplementedError("Dataset should expose .data / .data_ids (DynamicItemDataset protocol) when using length function")
            self._ex_lengths = [length_func(dataset.data[dataset.data_ids[i]]) for i in range(len(dataset))]
        if len(bucket_boundaries) > 0:
            if not all(x >= 0 for x in bucket_boundaries):
                raise ValueError("All elements in bucket boundaries should be non-negative (>= 0).")
            if len(set(bucket_boundaries)) != len(bucket_boundaries):
                raise ValueError("Bucket_boundaries should not contain duplicates.")
            if list(bucket_boundaries) != sorted(bucket_boundaries):
                raise AssertionError("The arg bucket_boundaries should be an ascending sorted list of non negative values values!")
            self._bucket_boundaries = np.array(sorted(bucket_boundaries))
        else:
            self._bucket_boundaries = np.array(self._warped_boundaries(max_batch_length, num_buckets))
        self._max_batch_length, self._shuffle_ex, self._batch_ordering = max_batch_length, shuffle, batch_ordering
        self._seed, self._drop_last, self._epoch, self.verbose = seed, drop_last, epoch, verbose
        self._max_batch_ex = np.inf if max_batch_ex is None else max_batch_ex
        self._bucket_lens = [max(1, int(max_batch_length / b)) for b in self._bucket_boundaries] + [1]
        self._generate_batches()

    @staticmethod
    def _warped_boundaries(max_batch_length, num_quantiles):
        """Right bucket edges = quantiles of a unit log-normal at k/(n+1), rescaled so that the last one is max_batch_length."""
        from scipy.stats import lognorm
        nb = num_quantiles + 1
        q = lognorm.ppf(np.linspace(1 / nb, num_quantiles / nb, num_quantiles), 1)
        return sorted(q * max_batch_length / q[-1])

    def get_durations(self, batch):
        return [self._ex_lengths[i] for i in batch]

    def _generate_batches(self):
        n = len(self._ex_lengths) if self._dataset is None else len(self._dataset)
        if self._shuffle_ex:
            g = torch.Generator()
            g.manual_seed(self._seed + self._epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = range(n)
        batches, open_ = [], [[] for _ in self._bucket_lens]
        for idx in order:
            b = int(np.searchsorted(self._bucket_boundaries, self._ex_lengths[idx]))
            open_[b].append(idx)
            if len(open_[b]) >= self._bucket_lens[b] or len(open_[b]) >= self._max_batch_ex:
                batches.append(open_[b])
                open_[b] = []
        if not self._drop_last:
            batches.extend(b for b in open_ if b)
        longest = lambda batch: max(self._ex_lengths[i] for i in batch)  # noqa: E731
        if self._batch_ordering == "random":
            g = torch.Generator()
            g.manual_seed(self._seed + self._epoch)
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        elif self._batch_ordering == "ascending":
            batches = sorted(batches, key=longest)
        elif self._batch_ordering == "descending":
            batches = sorted(batches, key=longest, reverse=True)
        else:
            raise NotImplementedError
        self._batches = batches

    def __iter__(self):
        for batch in self._batches:
            yield batch
        if self._shuffle_ex:            # re-bucket with the next epoch's permutation
            self._generate_batches()

    def set_epoch(self, epoch):
        self._epoch = epoch
        self._generate_batches()

    def __len__(self):
        return len(self._batches)


def load_manifest(json_path, replacements=None):
    """{utterance id: entry} with ``{key}`` placeholders in string fields substituted (e.g. {"data_folder": "/data/LibriSpeechMix"})."""
    with open(json_path) as f:
        data = json.load(f)
    if replacements:
        def sub(v):
            if isinstance(v, str):
                for k, r in replacements.items():
                    v = v.replace("{" + k + "}", r)
                return v
            if isinstance(v, list):
                return [sub(x) for x in v]
            if isinstance(v, dict):
                return {k: sub(x) for k, x in v.items()}
            return v
        data = sub(data)
    return data


def mix_sources(sigs, delays, start, duration, target_speaker_idx, sample_rate=16000, gain_nontarget=0):
    """The mixture of train_librispeechmix_scratch.py:356-386: every non-target source is rescaled so that its mean power is
    10^(gain_nontarget/10) times the target's (gain_nontarget == 0: sources are summed as they are), each source is shifted right by
    ceil(delay * sample_rate) samples, all are zero-padded to the longest and summed, and the window
    [ceil(start * sr), ceil(start * sr) + ceil(duration * sr)) is returned."""
    sigs = [s.clone() for s in sigs]
    out = []
    for i, (sig, delay) in enumerate(zip(sigs, delays)):
        if i != target_speaker_idx and gain_nontarget != 0:
            target_power = (sigs[target_speaker_idx] ** 2).mean()
            gain = (10 ** (gain_nontarget / 10) * target_power / (sig ** 2).mean()).sqrt()
            sig = sig * gain
        out.append(torch.nn.functional.pad(sig, [math.ceil(delay * sample_rate), 0]))
    n = max(len(x) for x in out)
    out = [torch.nn.functional.pad(x, [0, n - len(x)]) for x in out]
    mixed = out[0].clone()
    for x in out[1:]:          # left to right, as the reference adds them (:375-377): fp32 sums of 3+ sources depend on the order
        mixed += x
    a = math.ceil(start * sample_rate)
    return mixed[a:a + math.ceil(duration * sample_rate)]


MIX_MAX_SRC = 8          # csrc/dataio.hip


def mix_sources_device(sigs, delays, start, duration, target_speaker_idx, sample_rate=16000, gain_nontarget=0):
    """mix_sources on the GPU (csrc/dataio.hip, tsasr_mix_sources): ``sigs`` are 1-D fp32 device tensors; returns the cropped mixture as a
    device tensor. The host only converts seconds to samples with the reference's own ceil (train_librispeechmix_scratch.py:370,379-385).
    Bit-identical to mix_sources when gain_nontarget == 0; with a gain, the two mean powers behind it are fp64 sums rounded once (the
    reference's fp32 cascade sum depends on the host's SIMD width), every other operation is the reference's fp32 operation."""
    import ctypes
    from . import _capi as C
    C.require_gpu(*sigs)
    n = len(sigs)
    if n > MIX_MAX_SRC:      # the kernel takes up to 8 sources (LibriSpeechMix has 1 - 3): a longer list is mixed by the host formulas on the device tensors
        return mix_sources(list(sigs), delays, start, duration, target_speaker_idx, sample_rate, gain_nontarget)
    lens = [int(s.numel()) for s in sigs]
    src = sigs[0].float().contiguous() if n == 1 else torch.cat([s.float().reshape(-1) for s in sigs])
    off = (ctypes.c_longlong * (n + 1))(*([0] + [sum(lens[:j + 1]) for j in range(n)]))
    dly = (ctypes.c_int * n)(*[math.ceil(d * sample_rate) for d in delays])
    a, L = math.ceil(start * sample_rate), math.ceil(duration * sample_rate)
    out_len = int(C.lib().tsasr_mix_sources_out_len(off, dly, n, a, L))
    out = torch.empty(out_len, dtype=torch.float32, device=src.device)
    if out_len == 0:         # the window starts beyond the mixture: an empty slice, as in the reference
        return out
    ws = torch.empty(max(int(C.lib().tsasr_mix_sources_workspace_bytes()), 16), dtype=torch.uint8, device=src.device)
    ratio = float(torch.tensor(10 ** (gain_nontarget / 10), dtype=torch.float32)) if gain_nontarget != 0 else 1.0
    C.check(C.lib().tsasr_mix_sources(C.ptr(src), off, dly, n, int(target_speaker_idx), ratio, int(gain_nontarget != 0), a, L,
                                      C.ptr(out), C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_mix_sources")
    return out


def trim_enroll(enroll_sig, trim_enroll_seconds, sample_rate=16000):
    return enroll_sig[: math.ceil(trim_enroll_seconds * sample_rate)]


def collate(examples, blank_index=0):
    """examples: dicts with id, mixed_sig [L], enroll_sig [Le], tokens (list / 1-D int tensor) -> the PaddedBatch TSASR.fit_batch takes
    (tokens_bos = [blank] + tokens, relative lengths = len / max len, as speechbrain.dataio.batch.PaddedBatch)."""
    def pad(seqs, dtype=None):
        seqs = [torch.as_tensor(s) if dtype is None else torch.as_tensor(s, dtype=dtype) for s in seqs]
        n = max(int(s.shape[0]) for s in seqs)
        data = torch.stack([torch.nn.functional.pad(s, [0, n - s.shape[0]]) for s in seqs])
        return PaddedData(data, torch.tensor([s.shape[0] / n for s in seqs], dtype=torch.float32))
    toks = [torch.as_tensor(e["tokens"], dtype=torch.long) for e in examples]
    return PaddedBatch({
        "id": [e["id"] for e in examples],
        "mixed_sig": pad([e["mixed_sig"] for e in examples]),
        "enroll_sig": pad([e["enroll_sig"] for e in examples]),
        "tokens_bos": pad([torch.cat([torch.tensor([blank_index]), t]) for t in toks]),
        "tokens": pad(toks),
    })


def manifest_batches(json_path, hparams, data_folder=None, device="cpu"):
    """Batches for Brain.fit / evaluate from a LibriSpeechMix manifest (the JSON of librispeechmix_prepare.py:206-218). Decoding audio
    and tokenising text are the caller's job (no torchaudio / dataset on the GPU box): every entry names, next to the reference's keys,
    a `tensors` file (torch.save of {"sigs": [1-D waveforms of the sources], "enroll_sig": 1-D, "tokens": 1-D int}). The mixture is
    built here as audio_pipeline does (mix_sources: gains, delays, sum, crop; trim_enroll), examples are sorted by duration
    (`sorting: ascending`) and grouped by DynamicBatchSampler when hparams has `max_batch_length`, else in fixed `batch_size` groups.
    device="cuda" builds the mixtures with csrc/dataio.hip; with `gain_nontarget` != 0 those differ from device="cpu" in the last bits (the two
    mean powers are fp64 sums rounded once on the device, fp32 cascade sums on the host: a few ulp of the samples) - train and evaluate one
    manifest with one setting when runs must be comparable bit for bit."""
    entries = load_manifest(json_path, {"data_folder": data_folder} if data_folder else None)
    sr = int(hparams.get("sample_rate", 16000))
    on_device = torch.device(device).type == "cuda"      # the mixture is then built by the HIP kernel from the sources' device copies
    items = []
    for uid, e in entries.items():
        t = torch.load(e["tensors"])
        mix = mix_sources_device if on_device else mix_sources
        mixed = mix([s.float().to(device) if on_device else s.float() for s in t["sigs"]], e["delays"], e.get("start", 0.0), e["duration"],
                    e["target_speaker_idx"], sr, hparams.get("gain_nontarget", 0))
        enroll = trim_enroll(t["enroll_sig"].float().to(device) if on_device else t["enroll_sig"].float(), hparams.get("trim_enroll", 20.0), sr)
        items.append({"id": uid, "duration": float(e["duration"]), "mixed_sig": mixed, "enroll_sig": enroll, "tokens": t["tokens"]})
    items.sort(key=lambda x: x["duration"])
    if hparams.get("max_batch_length"):
        sampler = DynamicBatchSampler(items, hparams["max_batch_length"], num_buckets=hparams.get("num_buckets", 20), shuffle=False,
                                      batch_ordering="ascending", lengths_list=[x["duration"] for x in items])
        groups = [list(b) for b in sampler]
    else:
        bs = int(hparams.get("batch_size", 8))
        groups = [list(range(i, min(i + bs, len(items)))) for i in range(0, len(items), bs)]
    return [collate([items[i] for i in g], hparams.get("blank_index", 0)).to(device) for g in groups]
