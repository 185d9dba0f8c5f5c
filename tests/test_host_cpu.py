"""CPU-side tests of the host logic: C-ABI surface, mini HyperPyYAML, state_dict compatibility, schedulers, CLI,
Brain bookkeeping and the data-parallel gradient arena (gloo, world_size 2). No kernel is launched here."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    capi = importlib.import_module("ts-asr_amd._capi")
    header = open(os.path.join(ROOT, "include", "tsasr_hip.h")).read()
    declared = set(re.findall(r"\b(tsasr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/tsasr_hip.h but not exported by libtsasr_hip.so"
    assert declared == set(capi.exported_symbols()), declared ^ set(capi.exported_symbols())
    assert capi.lib().tsasr_version() >= 1


def test_product_ops_refuse_cpu_tensors(pkg):
    rn = importlib.import_module("ts-asr_amd.rnnt")
    ops = importlib.import_module("ts-asr_amd.ops")
    capi = importlib.import_module("ts-asr_amd._capi")
    with pytest.raises(capi.TsasrHipMissing):
        ops.layer_norm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-5)
    with pytest.raises(capi.TsasrHipMissing):
        rn.fused_joint_logits(torch.zeros(1, 2, 32), torch.zeros(1, 2, 32), torch.zeros(5, 32), torch.zeros(5))


def _load(path, ov=None):
    hp = importlib.import_module("ts-asr_amd.hparams")
    with open(path) as f:
        return hp.load_hyperpyyaml(f, ov)


SMALL = dict(d_model=144, nhead=4, encoder_num_layers=2, speaker_num_layers=2, d_ffn=576, joint_dim=160, decoder_neurons=128)


def test_yaml_loader_and_state_dict_keys_match_reference_shapes():
    from tests.test_oracle_golden import full_state_dict
    from oracle.golden_recipe import CFG1
    h = _load(os.path.join(ROOT, "hparams", "conformer-t_scratch_mi355x.yaml"), SMALL)
    assert h["decoder"].rnn.input_size == 28                       # !ref <vocab_size> - 1
    assert h["opt_class"].keywords == {"lr": 0.001, "betas": (0.9, 0.98), "eps": 1e-08, "weight_decay": 0.01}
    assert h["modules"]["encoder"] is h["encoder"]                 # !ref returns the same object
    ref = full_state_dict(CFG1, "cat")                             # key -> tensor, shapes as the reference's state_dict
    mine = {f"{n}.{k}": tuple(v.shape) for n, m in h["modules"].items() for k, v in m.state_dict().items()}
    for k, v in ref.items():
        assert mine.get(k) == tuple(v.shape), (k, mine.get(k), tuple(v.shape))
    extra = set(mine) - set(ref) - {"embedding.Embedding.weight", "encoder.positional_encoding.inv_freq",
                                    "speaker_encoder.positional_encoding.inv_freq"}
    assert not extra, extra


@pytest.mark.skipif(not os.path.isdir("/root/reference/hparams"), reason="reference YAMLs only exist in the build container")
@pytest.mark.parametrize("name", ["conformer-t_scratch", "conformer-t_wavlm", "conformer-t_none"])
def test_reference_yaml_files_load_unchanged(name):
    h = _load(f"/root/reference/hparams/LibriSpeechMix/{name}.yaml", {"data_folder": "/nonexistent"})
    n = sum(p.numel() for m in h["modules"].values() if isinstance(m, torch.nn.Module) for p in m.parameters() if p.requires_grad)
    assert n == {"conformer-t_scratch": 51022749}.get(name, n)     # trainable total of the scratch recipe
    assert type(h["augmentation"]).__name__ == "SpecAugment" and type(h["speed_perturb"]).__name__ == "SpeedPerturb"
    aug = h["augmentation"]               # conformer-t_scratch.yaml:132-142
    assert (aug.time_warp_window, aug.freq_mask_width, aug.time_mask_width, aug.replace_with_zero) == (5, (0, 30), (0, 20), False)
    assert [r.new_freq for r in h["speed_perturb"].resamplers] == [15200, 16000, 16800]
    assert type(h["train_logger"]).__name__ == "Unavailable"      # off the hot path: placeholder that raises only when used
    with pytest.raises(ValueError):
        _load(f"/root/reference/hparams/LibriSpeechMix/{name}.yaml")  # !PLACEHOLDER data_folder must be overridden


def test_noam_scheduler_and_epoch_counter():
    core = importlib.import_module("ts-asr_amd.core")
    from oracle.tsasr_ref import noam_lr
    sch = core.NoamScheduler(lr_initial=1e-3, n_warmup_steps=10000)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    for n in range(1, 6):
        _, lr = sch(opt)
        assert lr == pytest.approx(noam_lr(1e-3, n, 10000), rel=1e-12)
        assert opt.param_groups[0]["lr"] == lr
    sch.n_steps = 9999
    assert sch(opt)[1] == pytest.approx(1e-3, rel=1e-9)             # peak at n = warm-up
    assert list(core.EpochCounter(3)) == [1, 2, 3]


def test_parse_arguments_like_speechbrain(monkeypatch):
    core = importlib.import_module("ts-asr_amd.core")
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    f, run, ov = core.parse_arguments(["hp.yaml", "--device", "cuda:0", "--distributed_launch", "--injection_mode", "sum",
                                       "--causal_encoder", "True", "--max_grad_norm", "1.5", "--lr=0.01"])
    assert f == "hp.yaml" and run["distributed_launch"] is True and run["max_grad_norm"] == 1.5
    assert ov == {"injection_mode": "sum", "causal_encoder": True, "lr": 0.01}
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert core.parse_arguments(["hp.yaml"])[1]["device"] == "cuda:3"


def test_synthetic_batch_surface():
    bm = importlib.import_module("ts-asr_amd.batch")
    b = bm.synthetic_batch(4, 3200, 1600, 10, ragged=True)
    sig, lens = b.mixed_sig
    assert sig.shape == (4, 3200) and lens[-1] == 1.0 and torch.all(lens[:-1] <= lens[1:])
    assert b.tokens_bos.data.shape == (4, 11) and torch.all(b.tokens_bos.data[:, 0] == 0)
    assert torch.all(b.tokens.data[0, int(round(float(b.tokens.lengths[0]) * 10)):] == 0)


class _ToyBrain:
    pass


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    core = importlib.import_module("ts-asr_amd.core")
    dp = importlib.import_module("ts-asr_amd.dp")
    dp.ddp_init_group({"distributed_launch": True, "distributed_backend": "gloo"})
    torch.manual_seed(0)
    mods = {"a": torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16)),
            "b": torch.nn.Linear(16, 3)}
    if rank != 0:       # ranks start from DIFFERENT weights: the arena must adopt rank 0's (the reference's DDP constructor broadcast)
        with torch.no_grad():
            for m in mods.values():
                for p_ in m.parameters():
                    p_.add_(1.0)

    class Toy(core.Brain):
        def compute_forward(self, batch, stage):
            return self.modules.b(self.modules.a(batch[0]))

        def compute_objectives(self, predictions, batch, stage):
            return ((predictions - batch[1]) ** 2).mean()

    import functools
    brain = Toy(mods, functools.partial(torch.optim.SGD, lr=0.1), {}, {"device": "cpu", "distributed_launch": True,
                                                                     "distributed_backend": "gloo", "grad_accumulation_factor": 2,
                                                                     "max_grad_norm": 0.0})
    brain.arena_kwargs = {}
    g = torch.Generator().manual_seed(100)
    data = [(torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)) for _ in range(6)]
    for step, (x, y) in enumerate(data):   # each rank trains on its half of every batch
        brain.fit_batch((x[rank::world], y[rank::world]))
    torch.save({k: v.clone() for k, v in brain.modules.state_dict().items()}, os.path.join(out_dir, f"w{rank}.pt"))
    torch.save({"steps": brain.optimizer_step, "buckets": len(brain.arena.buckets),
                "order": [[id(q) for q in brain.arena.params_initial].index(id(p_)) for p_ in brain.arena.params_ordered]}, os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_arena_gloo_world2(tmp_path):
    """2 ranks x half batches with overlapped bucket all-reduce == 1 process on the full batches (grad accumulation 2)."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                          # ranks stay in lock-step
    assert torch.load(tmp_path / "m0.pt")["steps"] == 3               # 6 micro-batches / accumulation 2
    assert torch.load(tmp_path / "m0.pt")["order"] == torch.load(tmp_path / "m1.pt")["order"]   # every rank laid its arena out in rank 0's order
    # single-process reference on the full batches
    core = importlib.import_module("ts-asr_amd.core")
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16))
    b = torch.nn.Linear(16, 3)
    opt = torch.optim.SGD(list(a.parameters()) + list(b.parameters()), lr=0.1)
    g = torch.Generator().manual_seed(100)
    data = [(torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)) for _ in range(6)]
    for step, (x, y) in enumerate(data):
        # mean over the two half-batch losses == what the two ranks average
        loss = sum(((b(a(x[r::2])) - y[r::2]) ** 2).mean() for r in range(2)) / 2
        (loss / 2).backward()
        if step % 2 == 1:
            opt.step()
            opt.zero_grad()
    ref = {**{"a." + k: v for k, v in a.state_dict().items()}, **{"b." + k: v for k, v in b.state_dict().items()}}
    for k in ref:
        torch.testing.assert_close(w0[k], ref[k], atol=1e-6, rtol=1e-5)


def test_brain_counts_nonfinite_losses_on_cpu():
    core = importlib.import_module("ts-asr_amd.core")
    import functools

    class Toy(core.Brain):
        def compute_forward(self, batch, stage):
            return self.modules.m(batch)

        def compute_objectives(self, predictions, batch, stage):
            return predictions.sum() * float("nan")

    brain = Toy({"m": torch.nn.Linear(2, 2)}, functools.partial(torch.optim.SGD, lr=0.1), {"nonfinite_patience": 1}, {"device": "cpu"})
    brain.fit_batch(torch.ones(1, 2))
    assert brain.flush_nonfinite() == 1
    brain.fit_batch(torch.ones(1, 2))
    with pytest.raises(ValueError):
        brain.flush_nonfinite()


def test_resampler_host_side_matches_reference(golden):
    """Host logic of speed perturbation: the filter bank nnet.Resample builds equals the reference's (golden, bit for bit), the
    C-ABI's output-length rule equals the oracle's for ragged lengths, and SpeedPerturb follows torch's CPU generator exactly like
    the reference (tests/golden/c1_augment.npz: 64 draws after torch.manual_seed(11))."""
    from oracle import tsasr_ref as R
    nn_ = importlib.import_module("ts-asr_amd.nnet")
    capi = importlib.import_module("ts-asr_amd._capi")
    g = golden["c1_augment"]
    for speed in (95, 105, 50):
        rs = nn_.Resample(16000, 16000 * speed // 100)
        w, first = rs._filters(torch.device("cpu"))
        assert np.array_equal(w.numpy(), g[f"sp_{speed}_weights"]) and np.array_equal(first.numpy(), g[f"sp_{speed}_first"].astype(np.int32))
        for n in (0, 1, 19, 20, 21, 3999, 4000, 159840, 159999):
            assert capi.lib().tsasr_resample_out_len(n, 16000, rs.new_freq) == R.resample_out_len(n, 16000, rs.new_freq)
    sp = nn_.SpeedPerturb(16000, speeds=[95, 100, 105])
    picks = []
    torch.manual_seed(11)
    for _ in range(64):
        assert float(torch.rand(1)) <= 1.0                       # the reference's perturb_prob draw comes first
        picks.append(int(torch.randint(len(sp.speeds), (1,))[0]))
    assert picks == g["sp_index_draws"].tolist()
    no = nn_.SpeedPerturb(16000, perturb_prob=0.0)               # edge cases of the reference's unit test (test_augment.py:105-108)
    x = torch.sin(torch.arange(1600.0)).unsqueeze(0)
    assert torch.equal(no(x), x) and no(x) is not x
    assert nn_.SpeedPerturb(16000, speeds=[100])(x) is x


SAMPLER_CASES = {
    "train": dict(max_batch_length=50.0, num_buckets=80, shuffle=False, batch_ordering="ascending", max_batch_ex=None),
    "valid": dict(max_batch_length=50.0, num_buckets=80, shuffle=False, batch_ordering="descending", max_batch_ex=6),
    "shuffled_e0": dict(max_batch_length=30.0, num_buckets=12, shuffle=True, batch_ordering="random", seed=7, epoch=0),
    "shuffled_e3": dict(max_batch_length=30.0, num_buckets=12, shuffle=True, batch_ordering="random", seed=7, epoch=3),
    "boundaries": dict(max_batch_length=40.0, bucket_boundaries=[5.0, 10.0, 20.0, 40.0], shuffle=False, batch_ordering="descending",
                       drop_last=True),
}


@pytest.mark.parametrize("name", list(SAMPLER_CASES))
def test_dynamic_batch_sampler_equals_reference(golden, name):
    """Batch compositions of dataio.DynamicBatchSampler against the reference's own sampler run on the same 500 durations
    (tests/golden/c1_sampler.npz, oracle/gen_golden_sampler.py): the recipe's train / valid settings and the shuffled, random-order,
    max_batch_ex and explicit-boundary variants - index for index."""
    dataio = importlib.import_module("ts-asr_amd.dataio")
    g = golden["c1_sampler"]
    lens = g["durations"].tolist()
    s = dataio.DynamicBatchSampler(list(range(len(lens))), lengths_list=lens, **SAMPLER_CASES[name])
    assert np.allclose(s._bucket_boundaries, g[name + "_boundaries"], rtol=1e-12)
    batches = [list(b) for b in s]
    assert [len(b) for b in batches] == g[name + "_sizes"].tolist() and len(s) == len(batches)
    assert [i for b in batches for i in b] == g[name + "_flat"].tolist()
    kw = SAMPLER_CASES[name]
    if not kw.get("drop_last"):
        assert sorted(i for b in batches for i in b) == list(range(len(lens)))        # every utterance exactly once
    for b in batches:                                                                      # a bucket's batch never exceeds its budget
        edge = g[name + "_boundaries"][min(int(np.searchsorted(g[name + "_boundaries"], max(lens[i] for i in b))), len(g[name + "_boundaries"]) - 1)]
        assert len(b) <= max(1, int(kw["max_batch_length"] / edge)) or max(lens[i] for i in b) > g[name + "_boundaries"][-1]


def test_mixing_arithmetic_and_collate():
    """dataio.mix_sources on hand-computable cases (train_librispeechmix_scratch.py:356-386): delays in samples, zero padding, crop
    window, interference rescaled to the requested power ratio; collate -> PaddedBatch with relative lengths and blank-prefixed tokens."""
    dataio = importlib.import_module("ts-asr_amd.dataio")
    sr = 10
    a, b = torch.arange(1.0, 9.0), torch.ones(4) * 2
    m = dataio.mix_sources([a, b], delays=[0.0, 0.25], start=0.1, duration=0.6, target_speaker_idx=0, sample_rate=sr)
    full = torch.tensor([1, 2, 3, 4 + 2, 5 + 2, 6 + 2, 7 + 2, 8.0])          # b delayed by ceil(2.5) = 3 samples
    assert torch.equal(m, full[1:7])
    m2 = dataio.mix_sources([a, b], [0.0, 0.0], 0.0, 0.8, 0, sr, gain_nontarget=-10)
    interferer = m2 - a
    assert float((interferer[:4] ** 2).mean() / (a ** 2).mean()) == pytest.approx(0.1, rel=1e-5) and torch.all(interferer[4:] == 0)
    assert torch.equal(dataio.trim_enroll(torch.arange(100.0), 2.55, sr), torch.arange(26.0))
    # three sources: the sum runs left to right in fp32 like the reference's `mixed_sig += sig` loop (:375-377) - bitwise, not just close
    g = torch.Generator().manual_seed(0)
    s3 = [torch.randn(1000, generator=g) * 10 ** k for k in (0, 3, -3)]
    m3 = dataio.mix_sources(s3, [0.0, 0.0, 0.0], 0.0, 100.0, 1, sr)
    assert torch.equal(m3, (s3[0] + s3[1]) + s3[2])
    batch = dataio.collate([dict(id="u1", mixed_sig=torch.randn(50), enroll_sig=torch.randn(30), tokens=[3, 4, 5]),
                            dict(id="u2", mixed_sig=torch.randn(40), enroll_sig=torch.randn(15), tokens=[7])])
    assert batch.mixed_sig.data.shape == (2, 50) and batch.mixed_sig.lengths.tolist() == pytest.approx([1.0, 0.8])
    assert batch.tokens_bos.data.tolist() == [[0, 3, 4, 5], [0, 7, 0, 0]] and batch.tokens_bos.lengths.tolist() == pytest.approx([1.0, 0.5])
    assert batch.tokens.data.tolist() == [[3, 4, 5], [7, 0, 0]] and batch.id == ["u1", "u2"]


def test_mixing_equals_reference_audio_pipeline(golden):
    """dataio.mix_sources / trim_enroll against the REFERENCE's own `audio_pipeline` closure (train_librispeechmix_scratch.py:333-456, run
    by oracle/gen_golden_mix.py through speechbrain's DynamicItemDataset with only torchaudio.load / resample stubbed): two and three
    sources, gain_nontarget 0 / -5 / +3 dB, fractional delays, a crop that starts inside the mixture, enrollment trims. Bit for bit:
    the same fp32 operations in the same order (gain as a 0-dim tensor product, left-to-right sum)."""
    dataio = importlib.import_module("ts-asr_amd.dataio")
    g = golden["c1_mix"]
    cases = [str(c) for c in g["cases"]]
    assert len(cases) == 6
    for c in cases:
        start, duration, target, gain, trim, n = g[c + ".meta"].tolist()
        sigs = [torch.from_numpy(g[f"{c}.src{j}"].copy()) for j in range(int(n))]
        gain = int(gain) if float(gain).is_integer() else gain
        mixed = dataio.mix_sources(sigs, g[c + ".delays"].tolist(), start, duration, int(target), 16000, gain)
        ref = torch.from_numpy(g[c + ".mixed_sig"])
        assert mixed.shape == ref.shape and mixed.dtype == ref.dtype, c
        assert torch.equal(mixed, ref), (c, float((mixed - ref).abs().max()))
        enroll = dataio.trim_enroll(torch.from_numpy(g[c + ".enroll_src"].copy()), trim)
        assert torch.equal(enroll, torch.from_numpy(g[c + ".enroll_sig"])), c
        for j, s_ in enumerate(sigs):                      # the sources handed in are not modified (the reference scales its copies in place)
            assert torch.equal(s_, torch.from_numpy(g[f"{c}.src{j}"])), (c, j)


def test_manifest_loader(tmp_path):
    dataio = importlib.import_module("ts-asr_amd.dataio")
    entry = {"utt1": {"wavs": ["{data_folder}/a.flac", "{data_folder}/b.flac"], "enroll_wav": "{data_folder}/e.flac", "delays": [0.0, 1.5],
                      "start": 0.0, "duration": 10.0, "durations": [8.0, 8.5], "target_speaker_idx": 0, "wrd": "HELLO", "speakers": ["1", "2"],
                      "genders": ["m", "f"]}}
    p = tmp_path / "m.json"
    p.write_text(__import__("json").dumps(entry))
    d = dataio.load_manifest(str(p), {"data_folder": "/data"})
    assert d["utt1"]["wavs"] == ["/data/a.flac", "/data/b.flac"] and d["utt1"]["enroll_wav"] == "/data/e.flac" and d["utt1"]["delays"] == [0.0, 1.5]


def test_no_packed_fp32_instructions_in_device_code(tmp_path):
    """Device-code lint. On MI355X (ROCm 7.2) the packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, operand
    halves picked by op_sel) of one wave occasionally return wrong values while ANOTHER hardware queue's MFMA kernel runs beside it:
    round 1 met it as run-to-run different low bits of one `v_pk_add_f32 op_sel` sum, round 3 as one corrupted FFT frame of the log-mel
    kernel per ~40 launches beside the per-step LSTM kernels - in eager two-stream runs and in hipGraph replays alike, 0 of 128,000
    launches once the instructions are gone (profiles/r03_notes.md section 1). The library
    is therefore compiled with the subtarget feature off (csrc/Makefile, NOPK); this check disassembles every gfx950 code object of the
    built library and accepts no packed-fp32 arithmetic at all."""
    import glob
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "ts-asr_amd", "lib", "libtsasr_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        pytest.skip("needs the built library and llvm-objdump")
    work = shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(work)], cwd=tmp_path, check=True, capture_output=True)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950*"))
    assert objs, "no gfx950 code objects found in the library"
    bad = re.compile(r"\bv_pk_(fma|mul|add|mov)_(f32|b32)\b")
    n_mfma = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        n_mfma += len(re.findall(r"v_mfma_", asm))
        hits = [l.strip() for l in asm.splitlines() if bad.search(l)]
        assert not hits, f"{os.path.basename(o)}: {len(hits)} packed-fp32 instructions, e.g. {hits[:3]}"
    assert n_mfma > 0   # the disassembly really saw the kernels


def _dp_bf16_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dp = importlib.import_module("ts-asr_amd.dp")
    dp.ddp_init_group({"distributed_launch": True, "distributed_backend": "gloo"})
    torch.manual_seed(0)
    mods = torch.nn.ModuleDict({"a": torch.nn.Linear(64, 64), "b": torch.nn.Linear(64, 8)})
    out = {}
    for dtype in ("fp32", "bf16"):
        arena = dp.GradArena(mods, world_size=world, bucket_bytes=4096)
        arena.comm_dtype = dtype
        arena._order_final = True
        g = torch.Generator().manual_seed(5 + rank)
        x = torch.randn(16, 64, generator=g)
        arena.grads.zero_()
        arena.begin_backward(True)
        mods["b"](torch.tanh(mods["a"](x))).square().mean().backward()
        arena.finish_backward()
        out[dtype] = arena.grads.clone()
        out[dtype + "_sent"] = len(arena.sent_log)
        for h_ in arena._hooks:
            h_.remove()
    torch.save(out, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_bf16_allreduce_payload_gloo_world2(tmp_path):
    """TSASR_ALLREDUCE_DTYPE=bf16: buckets are rounded to bf16, averaged and written back into the fp32 arena - every rank ends with the
    same gradients, within bf16 rounding of the fp32-payload average; the same number of bucket collectives is issued."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_dp_bf16_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(g0["fp32"], g1["fp32"]) and torch.equal(g0["bf16"], g1["bf16"])
    assert g0["fp32_sent"] == g0["bf16_sent"] >= 2
    rel = float((g0["bf16"] - g0["fp32"]).norm() / g0["fp32"].norm())
    assert 0 < rel < 8e-3, rel


def test_manifest_batches_mix_and_bucket(tmp_path):
    """dataio.manifest_batches: a LibriSpeechMix-format manifest (librispeechmix_prepare.py:206-218 keys) -> mixed, trimmed, sorted, batched."""
    import json
    dataio = importlib.import_module("ts-asr_amd.dataio")
    g = torch.Generator().manual_seed(0)
    man = {}
    for i, dur in enumerate([0.5, 0.25, 0.4, 0.3]):
        n = int(dur * 16000)
        t = {"sigs": [torch.randn(n, generator=g), torch.randn(n // 2, generator=g)], "enroll_sig": torch.randn(8000, generator=g),
             "tokens": torch.randint(1, 29, (5 + i,), generator=g)}
        torch.save(t, tmp_path / f"u{i}.pt")
        man[f"u{i}"] = {"wavs": ["a.flac", "b.flac"], "enroll_wav": "e.flac", "delays": [0.0, 0.1], "start": 0.0, "duration": dur,
                        "durations": [dur, dur / 2], "target_speaker_idx": 0, "wrd": "x", "speakers": ["s1", "s2"], "genders": ["m", "f"],
                        "tensors": str(tmp_path / f"u{i}.pt")}
    (tmp_path / "train.json").write_text(json.dumps(man))
    batches = dataio.manifest_batches(str(tmp_path / "train.json"), {"batch_size": 2, "trim_enroll": 0.25, "blank_index": 0})
    assert [b.id for b in batches] == [["u1", "u3"], ["u2", "u0"]]                       # ascending duration
    b0 = batches[0]
    assert b0.mixed_sig.data.shape == (2, int(0.3 * 16000)) and b0.enroll_sig.data.shape == (2, 4000)   # enrollment trimmed to 0.25 s
    assert torch.allclose(b0.mixed_sig.lengths, torch.tensor([4000 / 4800, 1.0]))
    t1 = torch.load(tmp_path / "u1.pt")
    ref = t1["sigs"][0].clone()
    ref[1600:3600] += t1["sigs"][1]                                                         # second source (2000 samples) delayed by 0.1 s
    assert torch.allclose(b0.mixed_sig.data[0, :4000], ref)
    assert b0.tokens_bos.data[0, 0] == 0 and torch.equal(b0.tokens_bos.data[0, 1:7], t1["tokens"])


def _src_index(o, k, n, mode):      # the padding rule of the front-end convolutions (SB/nnet/CNN.py:629-711: reflect / causal), as csrc/frontend.hip
    if mode == 1:
        i = 2 * o + k - 2
        return -1 if i < 0 else i
    i = 2 * o + k - 1
    if mode == 0:
        if i < 0:
            i = -i
        if i >= n:
            i = 2 * (n - 1) - i
        return i
    return -1 if (i < 0 or i >= n) else i


@pytest.mark.parametrize("B,T,F,causal", [(2, 37, 21, False), (2, 37, 21, True), (1, 500, 40, False), (3, 10, 8, True), (1, 2, 2, False), (2, 3, 3, False),
                                          (1, 4, 5, True), (2, 2, 7, True), (1, 6, 2, False)])
def test_conv_dgrad_plan_covers_every_pixel_with_its_taps(pkg, B, T, F, causal):
    """The host plan of the front-end block-2 data gradient (csrc/gemm.hip conv_dgrad_plan: classes of input pixels x tap slots, launch order of
    the 128-pixel tiles) against the definition: every input pixel lies in exactly one class row, its slots name exactly the (tap, output
    position) pairs whose padded source index is that pixel (the convolution's transpose, SB/nnet/CNN.py:629-711), the 1x1 branch rides where
    the centre tap does, and every tile is launched exactly once. Pure host code: no kernel runs."""
    capi = importlib.import_module("ts-asr_amd._capi")
    L = capi.lib()
    n = L.tsasr_conv3x3s2_dgrad_plan_bytes(B, T, F, int(causal))
    assert n > 0 and n % 4 == 0
    plan = np.zeros(n // 4, dtype=np.int32)
    assert L.tsasr_conv3x3s2_dgrad_plan(B, T, F, int(causal), plan.ctypes.data, n) == 0
    ncls, slots, coff, toff = (int(v) for v in plan[:4])
    To, Fo = (T - 1) // 2 + 1, (F - 1) // 2 + 1
    tm, fm = (1, 2) if causal else (0, 0)
    ckt, ckf = (2, 1) if causal else (1, 1)
    want_t = {i: sorted((k, o) for o in range(To) for k in range(3) if _src_index(o, k, T, tm) == i) for i in range(T)}
    want_f = {i: sorted((k, o) for o in range(Fo) for k in range(3) if _src_index(o, k, F, fm) == i) for i in range(F)}
    seen = np.zeros((B, T, F), dtype=np.int32)
    tiles_of = []
    for ci in range(ncls):
        c = plan[coff + 32 * ci: coff + 32 * ci + 32]
        rows, nt, nf, t0, tstep, ost, f0, fstep, osf, nst, nsf, res_st, res_sf = (int(v) for v in c[:13])
        kt, ct, kf, cf = c[16:20], c[20:24], c[24:28], c[28:32]
        assert rows == B * nt * nf and 1 <= nst <= 4 and 1 <= nsf <= 4
        tiles_of.append((rows + 127) // 128)
        for it in range(nt):
            ti = t0 + tstep * it
            got_t = sorted((int(kt[s]), it * ost + int(ct[s])) for s in range(nst) if 0 <= it * ost + int(ct[s]) < To)
            assert got_t == want_t[ti], (ti, got_t, want_t[ti])
        for jf in range(nf):
            fi = f0 + fstep * jf
            got_f = sorted((int(kf[s]), jf * osf + int(cf[s])) for s in range(nsf) if 0 <= jf * osf + int(cf[s]) < Fo)
            assert got_f == want_f[fi], (fi, got_f, want_f[fi])
        has_centre = [(s, q) for s in range(nst) for q in range(nsf) if kt[s] == ckt and kf[q] == ckf]
        assert ([(res_st, res_sf)] if res_st >= 0 else []) == has_centre
        if res_st >= 0:      # the centre tap reads x[2 t', 2 f']: the 1x1 branch's only source
            assert tstep in (0, 2) and (t0 % 2 == 0) and (f0 % 2 == 0) and 2 * int(ct[res_st]) == t0 and 2 * int(cf[res_sf]) == f0
        seen[:, t0: t0 + max(tstep, 1) * nt: max(tstep, 1), f0: f0 + max(fstep, 1) * nf: max(fstep, 1)] += 1
    assert (seen == 1).all()
    order = plan[toff: toff + 2 * slots].reshape(slots, 2)
    assert slots % 8 == 0 and slots >= sum(tiles_of)
    launched = sorted((int(a), int(b)) for a, b in order if a >= 0)
    assert launched == [(ci, t) for ci in range(ncls) for t in range(tiles_of[ci])]
