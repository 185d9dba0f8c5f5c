"""GPU parity of the data side: the LibriSpeechMix mixture built by the HIP kernel (csrc/dataio.hip,
tsasr_mix_sources) against mixtures the REFERENCE's own `audio_pipeline` produced (train_librispeechmix_scratch.py:333-456, run by
oracle/gen_golden_mix.py -> tests/golden/c1_mix.npz), and dataio.manifest_batches feeding the training step from device-side mixing."""
import importlib
import json
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def dataio():
    return importlib.import_module("ts-asr_amd.dataio")


@pytest.fixture(scope="module")
def mixgold():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_mix.npz"))


def _case(g, c):
    start, duration, target, gain, trim, n = g[c + ".meta"].tolist()
    sigs = [torch.from_numpy(g[f"{c}.src{j}"].copy()) for j in range(int(n))]
    gain = int(gain) if float(gain).is_integer() else gain
    return sigs, g[c + ".delays"].tolist(), start, duration, int(target), gain, trim


def test_device_mixture_equals_reference_audio_pipeline(dataio, mixgold):
    """Six mixtures of the reference's audio_pipeline (two / three sources, gain_nontarget 0 / -5 / +3 dB, fractional delays, a crop that
    starts inside the mixture). gain 0: BIT-EXACT (shift, zero padding, left-to-right fp32 sum, crop are the reference's operations).
    With a gain: the kernel forms sqrt(ratio * p_target / p_j) with the reference's fp32 operations, but p = mean(x^2) is an fp64 sum rounded
    once where torch's CPU sum is an fp32 cascade whose grouping depends on the host's SIMD width - the gain may differ by one ulp, so
    the samples are compared to 4 ulp of the summands (5e-7 relative + 3e-7 absolute). (Observed on the GPU box: its host CPU's torch.mean
    differs from the build container's in the last bit on one of these sources - the reference's own arithmetic is not bit-stable across hosts.)"""
    cases = [str(c) for c in mixgold["cases"]]
    assert len(cases) == 6
    exact = 0
    for c in cases:
        sigs, delays, start, duration, target, gain, _ = _case(mixgold, c)
        ref = torch.from_numpy(mixgold[c + ".mixed_sig"])
        out = dataio.mix_sources_device([s.to(DEV) for s in sigs], delays, start, duration, target, 16000, gain)
        torch.cuda.synchronize()
        assert out.shape == ref.shape and out.dtype == ref.dtype, c
        if gain == 0:
            assert torch.equal(out.cpu(), ref), (c, float((out.cpu() - ref).abs().max()))
            exact += 1
        else:
            np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=5e-7, atol=3e-7, err_msg=c)    # (atol: sums of rescaled sources cancel)
    assert exact >= 2


def test_device_mixture_edge_cases(dataio):
    """One source; a window beyond the mixture (empty result, as a Python slice); a window that ends past the mixture (clipped); unaligned
    output lengths (scalar tail stores); eight sources, nine (more than the kernel takes: the host formulas, same result); equality with the host mirror (itself bit-exact against
    the reference)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1001, generator=g)
    one = dataio.mix_sources_device([a.to(DEV)], [0.0], 0.0, 1.0, 0, 1000)
    assert torch.equal(one.cpu(), a[:1000])
    assert dataio.mix_sources_device([a.to(DEV)], [0.0], 2.0, 1.0, 0, 1000).numel() == 0
    clipped = dataio.mix_sources_device([a.to(DEV)], [0.0103], 0.5, 1.0, 0, 1000)
    host = dataio.mix_sources([a], [0.0103], 0.5, 1.0, 0, 1000)
    assert clipped.shape == host.shape and torch.equal(clipped.cpu(), host)
    many = [torch.randn(200 + 37 * j, generator=g) for j in range(8)]
    delays = [0.001 * j * j for j in range(8)]
    dev = dataio.mix_sources_device([m.to(DEV) for m in many], delays, 0.0131, 0.4007, 3, 1000)
    host = dataio.mix_sources(many, delays, 0.0131, 0.4007, 3, 1000)
    assert torch.equal(dev.cpu(), host)
    nine = dataio.mix_sources_device([m.to(DEV) for m in many] + [a.to(DEV)], delays + [0.0], 0.0, 1.0, 0, 1000)    # nine sources: beyond the kernel's
    assert torch.equal(nine.cpu(), dataio.mix_sources(many + [a], delays + [0.0], 0.0, 1.0, 0, 1000))              # eight, the host formulas take it


def test_manifest_batches_mix_on_the_device(dataio, tmp_path):
    """dataio.manifest_batches(device=cuda): the mixtures come out of the HIP kernel (gain_nontarget 0 here: bit-identical to the host
    pipeline), batches are sorted / bucketed / padded exactly as on the host, and a batch feeds TSASR.fit_batch."""
    g = torch.Generator().manual_seed(0)
    man = {}
    for i, dur in enumerate([0.5, 0.25, 0.4, 0.3]):
        n = int(dur * 16000)
        t = {"sigs": [torch.randn(n, generator=g) * 0.1, torch.randn(n // 2, generator=g) * 0.1], "enroll_sig": torch.randn(8000, generator=g) * 0.1,
             "tokens": torch.randint(1, 29, (5 + i,), generator=g)}
        torch.save(t, tmp_path / f"u{i}.pt")
        man[f"u{i}"] = {"wavs": ["a.flac", "b.flac"], "enroll_wav": "e.flac", "delays": [0.0, 0.1], "start": 0.0, "duration": dur,
                        "durations": [dur, dur / 2], "target_speaker_idx": 0, "wrd": "x", "speakers": ["s1", "s2"], "genders": ["m", "f"],
                        "tensors": str(tmp_path / f"u{i}.pt")}
    (tmp_path / "train.json").write_text(json.dumps(man))
    hp = {"batch_size": 2, "trim_enroll": 0.25, "blank_index": 0}
    host = dataio.manifest_batches(str(tmp_path / "train.json"), hp)
    dev = dataio.manifest_batches(str(tmp_path / "train.json"), hp, device=DEV)
    assert [b.id for b in dev] == [b.id for b in host]
    for bd, bh in zip(dev, host):
        assert bd.mixed_sig.data.is_cuda
        assert torch.equal(bd.mixed_sig.data.cpu(), bh.mixed_sig.data) and torch.equal(bd.mixed_sig.lengths.cpu(), bh.mixed_sig.lengths)
        assert torch.equal(bd.enroll_sig.data.cpu(), bh.enroll_sig.data) and torch.equal(bd.tokens_bos.data.cpu(), bh.tokens_bos.data)
    # one training step straight from the device-built batch (configs[0] model)
    entry = importlib.import_module("__graft_entry__")
    brain, _ = entry._config1_brain()
    brain.modules.train()
    loss = brain.fit_batch(dev[0])
    torch.cuda.synchronize()
    assert math.isfinite(float(loss))
