#!/usr/bin/env python3
"""Recipe entry point on the MI355X runtime - the counterpart of the reference's train_librispeechmix_{scratch,pretrained,none}.py
`__main__` (train_librispeechmix_scratch.py:491-611): parse_arguments -> load_hyperpyyaml -> ddp_init_group -> TSASR -> fit -> evaluate.

    python train_tsasr.py hparams/conformer-t_scratch_mi355x.yaml [--key value ...]            (one GPU)
    python -m torch.distributed.run --nproc-per-node N train_tsasr.py <yaml> --distributed_launch   (one process per GPU, RCCL)

Run options and YAML overrides are those of speechbrain.core.parse_arguments (SB/core.py:134-393); extra keys understood here:
  --train_json / --valid_json / --test_json : LibriSpeechMix manifests written by the reference's librispeechmix_prepare.py
        (+ --data_folder). Audio decoding and the SentencePiece tokenizer are the caller's (dataio.py docstring): a manifest entry
        must carry `sig_path` tensors (torch .pt with mixed_sig / enroll_sig / tokens) - see ts-asr_amd/dataio.py.
  --synthetic N    : N synthetic LibriSpeechMix-shaped batches per epoch instead of manifests (no dataset on the GPU box); shapes from
        --syn_batch / --syn_seconds / --syn_enroll_seconds / --syn_tokens; lengths are length-bucketed like `sorting: ascending`.
  --hip_graph True : capture the step into hipGraphs (one per batch shape).
The pretrained-speaker variant is picked from the YAML (conformer-t_wavlm_mi355x.yaml); batches then carry `enroll_emb`."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "ts-asr_amd"
core = importlib.import_module(PKG + ".core")
dp = importlib.import_module(PKG + ".dp")
hp_mod = importlib.import_module(PKG + ".hparams")
batch_mod = importlib.import_module(PKG + ".batch")
tsasr = importlib.import_module(PKG + ".recipes.tsasr")

EXTRA = {"synthetic": 0, "syn_batch": 8, "syn_seconds": 4.0, "syn_enroll_seconds": 2.0, "syn_tokens": 24, "hip_graph": False,
         "number_of_epochs": 1, "train_json": None, "valid_json": None, "test_json": None, "data_folder": None}


def synthetic_loader(n_batches, hparams, opts, seed, device):
    """n_batches seeded LibriSpeechMix-shaped batches; three length buckets (x 1.0 / 0.75 / 0.5) visited in ascending order."""
    sr = hparams.get("sample_rate", 16000)
    feats = bool(hparams.get("input_is_feats", False))
    emb = int(hparams.get("speaker_embedding_dim", 0)) if "speaker_encoder_path" in hparams else 0
    out = []
    for i in range(n_batches):
        scale = (0.5, 0.75, 1.0)[min(2, i * 3 // max(n_batches, 1))]
        n_mix = int(opts["syn_seconds"] * scale * (100 if feats else sr))
        n_enr = int(opts["syn_enroll_seconds"] * (100 if feats else sr))
        if not feats:   # Fbank frames = 1 + L // 160: keep L = 160 * frames - 160 so that T' is a round number
            n_mix, n_enr = n_mix // 160 * 160 - 160, n_enr // 160 * 160 - 160
        b = batch_mod.synthetic_batch(opts["syn_batch"], n_mix, n_enr, max(1, int(opts["syn_tokens"] * scale)), vocab_size=hparams.get("vocab_size", 29),
                                      seed=seed + i, ragged=True, feats=feats, enroll_emb_dim=emb)
        out.append(b.to(device))
    return out


def main(argv=None):
    hparams_file, run_opts, overrides = core.parse_arguments(argv)
    opts = {k: overrides.pop(k, v) for k, v in EXTRA.items()}
    with open(hparams_file) as f:
        hparams = hp_mod.load_hyperpyyaml(f, overrides)
    dp.ddp_init_group(run_opts)                                           # one process per GPU (SB/utils/distributed.py:123-201)
    brain = tsasr.TSASR(hparams["modules"], hparams["opt_class"], hparams, run_opts)
    if opts["hip_graph"]:
        brain.enable_hip_graph()
    rank = int(os.environ.get("RANK", 0))
    if opts["synthetic"]:
        train = synthetic_loader(int(opts["synthetic"]), hparams, opts, 1234 + 1000 * rank, brain.device)
        valid = synthetic_loader(max(1, int(opts["synthetic"]) // 4), hparams, opts, 99, brain.device)
        test = valid
    else:
        dataio = importlib.import_module(PKG + ".dataio")
        if not opts["train_json"]:
            raise SystemExit("give --train_json (a LibriSpeechMix manifest) or --synthetic N")
        load = lambda p: dataio.manifest_batches(p, hparams, opts["data_folder"], brain.device) if p else None  # noqa: E731
        train, valid, test = load(opts["train_json"]), load(opts["valid_json"]), load(opts["test_json"])
    hparams["epoch_counter"].limit = int(opts["number_of_epochs"])
    brain.fit(hparams["epoch_counter"], train, valid)
    result = {"train_loss": brain.avg_train_loss, "optimizer_steps": brain.optimizer_step, "nonfinite": brain.nonfinite_count}
    if test is not None:
        result["test_loss"] = brain.evaluate(test)
        result["test_hyps"] = getattr(brain, "last_hyps", None)
    if rank == 0:
        print({k: (v if k != "test_hyps" else (v[:2] if v else v)) for k, v in result.items()})
    if dp.is_initialized():
        torch.distributed.destroy_process_group()
    return brain, result


if __name__ == "__main__":
    main()
