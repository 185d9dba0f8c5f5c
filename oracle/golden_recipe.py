"""Shared recipe for the golden vectors (TEST INFRASTRUCTURE - never imported by the product path).

Both ``oracle/gen_golden.py`` (which imports the reference in the build container and dumps
``tests/golden/*.npz``) and the tests (which re-create the very same inputs and weights and run
the oracle / the HIP path on them) use this module, so neither inputs nor weights have to be
committed: they are a deterministic function of a name, generated with numpy's PCG64
(``numpy.random.default_rng``), whose stream is stable across numpy versions.

Shapes are BASELINE.json ``configs[0]``: 2-layer d_model=144 Conformer-T, 4 utterances of
T=200 mel frames (-> T'=50 encoder frames), U=20 tokens.
"""
import re
import zlib

import numpy as np

# ----------------------------------------------------------------------------------------------
# config 1 (BASELINE.json configs[0]); constructor arguments mirror
# /root/reference/hparams/LibriSpeechMix/conformer-t_scratch.yaml:121-245 with the reduced sizes.
# ----------------------------------------------------------------------------------------------
CFG1 = dict(
    sample_rate=16000, n_fft=512, n_mels=80, win_length=32,
    d_model=144, nhead=4, encoder_num_layers=2, speaker_num_layers=2, d_ffn=576,
    kernel_size=31, joint_dim=160, decoder_neurons=128, vocab_size=29, blank_index=0,
    frontend_channels=(128, 128), encoder_input_size=2560,
    B=4, L_mix=31840, L_enroll=15840, U=20,
    mix_lens=(1.0, 0.9, 0.8, 0.7), enroll_lens=(1.0, 0.75, 1.0, 0.5), tok_lens=(1.0, 0.9, 0.75, 0.5),
)

# Full-WIDTH model (BASELINE.json configs[1] layer shapes: d_model 256, Dh = 64, d_ffn 2048, joint 640, predictor 512) at two layers and two
# short utterances: the shapes the benchmarked kernels are built for (Dh = 64 attention, J = 640 joint, H = 512 LSTM), checked whole-model
# against the reference itself (oracle/gen_golden_d256.py -> tests/golden/c2_fullwidth.npz), not only against the oracle.
CFG2 = dict(
    sample_rate=16000, n_fft=512, n_mels=80, win_length=32,
    d_model=256, nhead=4, encoder_num_layers=2, speaker_num_layers=2, d_ffn=2048,
    kernel_size=31, joint_dim=640, decoder_neurons=512, vocab_size=29, blank_index=0,
    frontend_channels=(128, 128), encoder_input_size=2560,
    B=2, L_mix=31840, L_enroll=15840, U=20,
    mix_lens=(1.0, 0.8), enroll_lens=(1.0, 0.75), tok_lens=(1.0, 0.75),
)

_LN_PAT = re.compile(
    r"(\.norm\.(weight|bias)$)|(layer_norm\.(weight|bias)$)|(after_conv\.0\.(weight|bias)$)"
    r"|(ffn_module[12]\.0\.(weight|bias)$)|(norm[12]\.norm\.(weight|bias)$)"
)
_SKIP_PAT = re.compile(r"(inv_freq$)|(Embedding\.weight$)")


def _rng(name: str) -> np.random.Generator:
    return np.random.default_rng(zlib.crc32(name.encode()))


def det_tensor(name: str, shape, scale=1.0) -> np.ndarray:
    """float32 N(0, scale^2) tensor that depends only on (name, shape)."""
    return (_rng(name).standard_normal(tuple(shape)) * scale).astype(np.float32)


def det_weight(key: str, shape) -> "np.ndarray | None":
    """Deterministic parameter value for state_dict entry ``key``; None = leave the module's own."""
    if _SKIP_PAT.search(key):
        return None
    shape = tuple(shape)
    if _LN_PAT.search(key):
        base = 1.0 if key.endswith("weight") else 0.0
        return (base + det_tensor(key, shape, 0.1)).astype(np.float32)
    if "pos_bias_" in key:
        return det_tensor(key, shape, 0.2)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return det_tensor(key, shape, 1.0 / np.sqrt(fan_in))
    return det_tensor(key, shape, 0.1)


def det_state_dict(prefix: str, shapes: dict) -> dict:
    """{key: ndarray} for every key of ``shapes`` ({key: shape}) that is not skipped."""
    out = {}
    for k, shp in shapes.items():
        w = det_weight(prefix + k, shp)
        if w is not None:
            out[k] = w
    return out


def load_det_weights(module, prefix: str):
    """Overwrite every parameter/buffer of a torch module with its deterministic value."""
    import torch

    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        w = det_weight(prefix + k, v.shape)
        new[k] = v if w is None else torch.from_numpy(w).to(v.dtype)
    module.load_state_dict(new)
    return module


def golden_inputs(cfg=CFG1):
    """Synthetic LibriSpeechMix-shaped batch (numpy). Padded tails are zero, as PaddedBatch makes them."""
    B = cfg["B"]
    mix = det_tensor("in.mixed_sig", (B, cfg["L_mix"]), 0.1)
    enr = det_tensor("in.enroll_sig", (B, cfg["L_enroll"]), 0.1)
    mix_lens = np.asarray(cfg["mix_lens"], np.float32)
    enr_lens = np.asarray(cfg["enroll_lens"], np.float32)
    tok_lens = np.asarray(cfg["tok_lens"], np.float32)
    for b in range(B):
        mix[b, int(round(float(mix_lens[b]) * cfg["L_mix"])):] = 0.0
        enr[b, int(round(float(enr_lens[b]) * cfg["L_enroll"])):] = 0.0
    tokens = _rng("in.tokens").integers(1, cfg["vocab_size"], size=(B, cfg["U"])).astype(np.int64)
    for b in range(B):
        tokens[b, int(round(float(tok_lens[b]) * cfg["U"])):] = 0
    tokens_bos = np.concatenate([np.zeros((B, 1), np.int64), tokens], axis=1)
    # relative lengths of tokens_bos as the reference's dataio makes them: (len+1)/(U+1)
    tok_abs = np.round(tok_lens * cfg["U"]).astype(np.int64)
    tokens_bos_lens = ((tok_abs + 1) / float(cfg["U"] + 1)).astype(np.float32)
    return dict(
        mixed_sig=mix, mixed_lens=mix_lens, enroll_sig=enr, enroll_lens=enr_lens,
        tokens=tokens, tokens_lens=tok_lens, tokens_bos=tokens_bos, tokens_bos_lens=tokens_bos_lens,
    )


SPEAKER_EMBEDDING_DIM = 512   # hparams/LibriSpeechMix/conformer-t_wavlm.yaml:123 (microsoft/wavlm-base-sv x-vector size)


def golden_enroll_emb(cfg=CFG1):
    """Stand-in for the frozen WavLM x-vector of train_librispeechmix_pretrained.py:45-63: [B, 1, 512] N(0,1)."""
    return det_tensor("in.enroll_emb", (cfg["B"], 1, SPEAKER_EMBEDDING_DIM), 1.0)
