"""Host-side mirror of the reference modules on the TS-ASR hot path.

Every class keeps the reference's constructor arguments, call signature and ``state_dict`` key names/shapes so
that (i) the three hparams YAMLs work with only the ``!new:`` paths re-pointed and (ii) checkpoints written by
the reference load unchanged. What runs underneath is different: tensors stay on the MI355X, lengths stay on
the device (no ``.item()`` / ``.cpu()`` in the step - it must be hipGraph-capturable), activations are kept in
the *compute dtype* (bf16 by default), and the hot operators go through the C-ABI of libtsasr_hip.so
(``ops.py``). Nothing here runs on the CPU: see ``_capi.require_gpu``.

Reference files mirrored (SB = vendor/speechbrain/speechbrain):
  Linear            SB/nnet/linear.py:15-78            Embedding      SB/nnet/embedding.py:14-114
  LSTM              SB/nnet/RNN.py:170-278             LayerNorm      SB/nnet/normalization.py:172-223
  Fbank             SB/lobes/features.py:22-147        InputNormalization SB/processing/features.py:933-1216
  ConvolutionFrontEnd SB/lobes/models/convolution.py:103-266  (Conv2d: SB/nnet/CNN.py:513-739)
  RelPosEncXL / RelPosMHAXL / PositionalwiseFeedForward  SB/nnet/attention.py:312-359 / 362-639 / 778-836
  ConvolutionModule / ConformerEncoderLayer              SB/lobes/models/transformer/Conformer.py:24-115 / 118-260
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _capi as C
from . import ops

_COMPUTE_DTYPE = torch.bfloat16


def set_compute_dtype(dtype):
    """torch.bfloat16 (default, BASELINE configs[1]) or torch.float32 (parity runs)."""
    global _COMPUTE_DTYPE
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _COMPUTE_DTYPE = dtype


def compute_dtype():
    return _COMPUTE_DTYPE


def _cd(x):
    return x if x.dtype == _COMPUTE_DTYPE else x.to(_COMPUTE_DTYPE)


def abs_lengths_round(rel, dim):
    """(rel * dim).round() kept on the device (models/conformer.py:272, SB/nnet/losses.py:58-59): one HIP launch
    (round half to even like torch.round) instead of mul + round + cast."""
    if rel.is_cuda and rel.shape[0] <= 1024:
        return ops.abs_lengths(rel, dim, 0)
    return (rel * dim).round().to(torch.int32)


# ------------------------------------------------------------------------------------------------------
class Linear(nn.Module):
    def __init__(self, n_neurons, input_shape=None, input_size=None, bias=True, combine_dims=False):
        super().__init__()
        self.combine_dims = combine_dims
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4 and self.combine_dims:
                input_size = input_shape[2] * input_shape[3]
        self.w = nn.Linear(input_size, n_neurons, bias=bias)

    def forward(self, x):
        C.require_gpu(x)
        if x.ndim == 4 and self.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        return ops.linear(_cd(x), self.w.weight, self.w.bias)


class Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim=128, consider_as_one_hot=False, blank_id=0):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.consider_as_one_hot = consider_as_one_hot
        self.embedding_dim = num_embeddings - 1 if consider_as_one_hot else embedding_dim
        self.blank_id = blank_id
        if consider_as_one_hot:
            self.Embedding = nn.Embedding(num_embeddings, self.embedding_dim, padding_idx=blank_id)
            eye = torch.eye(self.embedding_dim)
            with torch.no_grad():
                self.Embedding.weight.zero_()
                self.Embedding.weight[blank_id + 1:] = eye[blank_id:]
                if blank_id != 0:
                    self.Embedding.weight[:blank_id] = eye[:blank_id]
            self.Embedding.weight.requires_grad = False
        else:
            self.Embedding = nn.Embedding(num_embeddings, self.embedding_dim)

    def forward(self, x):
        C.require_gpu(x)
        return _cd(F.embedding(x.long(), self.Embedding.weight))


class LSTM(nn.Module):
    """Predictor network. cuDNN-style packed sequences are replaced by masking: the recurrence runs over the padded
    batch and outputs beyond each length are zeroed, which equals pad_packed_sequence(pack_padded_sequence(..))
    on every valid and padded position (lengths = floor(rel*T), SB/nnet/RNN.py:35) without a host sync."""

    def __init__(self, hidden_size, input_shape=None, input_size=None, num_layers=1, bias=True, dropout=0.0,
                 re_init=True, bidirectional=False):
        super().__init__()
        self.reshape = False
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size.")
        if input_size is None:
            if len(input_shape) > 3:
                self.reshape = True
            input_size = int(torch.prod(torch.tensor(input_shape[2:])).item())
        if bidirectional:
            raise NotImplementedError("the TS-ASR predictor is unidirectional")
        self.rnn = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, dropout=dropout,
                           bidirectional=False, bias=bias, batch_first=True)
        if re_init:
            for name, p in self.rnn.named_parameters():
                if "weight_hh" in name:
                    nn.init.orthogonal_(p)

    def forward(self, x, hx=None, lengths=None):
        C.require_gpu(x)
        if self.reshape and x.ndim == 4:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        out, hn = ops.lstm(_cd(x), self.rnn, hx)
        if lengths is not None:     # zero the outputs beyond floor(len * T) (pad_packed_sequence's padding): one masking launch each way
            out = ops.mask_time(out, ops.abs_lengths(lengths, x.shape[1], 1))
        return out, hn

    def forward_tokens(self, tokens, embedding, lengths=None):
        """decoder(embedding(tokens), lengths=lengths) for the recipes' predictor (train_librispeechmix_scratch.py:117-119). With the
        frozen one-hot Embedding of the TS-ASR YAMLs and bf16 compute, the embedding lookup and the LSTM's input projection collapse
        into one launch (ops.lstm_onehot: a token selects a column of W_ih); anything else takes the two modules as they are."""
        if (_COMPUTE_DTYPE == torch.bfloat16 and isinstance(embedding, Embedding) and embedding.consider_as_one_hot
                and not embedding.Embedding.weight.requires_grad
                and ops.lstm_onehot_supported(tokens, self.rnn, embedding.num_embeddings)):
            C.require_gpu(tokens)
            out = ops.lstm_onehot(tokens, self.rnn, embedding.blank_id)
            if lengths is not None:
                out = ops.mask_time(out, ops.abs_lengths(lengths, tokens.shape[1], 1))
            return out, None
        return self.forward(embedding(tokens), lengths=lengths)


class LayerNorm(nn.Module):
    def __init__(self, input_size=None, input_shape=None, eps=1e-05, elementwise_affine=True):
        super().__init__()
        self.eps = eps
        if input_shape is not None:
            input_size = input_shape[2:]
        self.norm = nn.LayerNorm(input_size, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, x):
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.eps)


# ------------------------------------------------------------------------------------------------------
# A1 / A2 features
# ------------------------------------------------------------------------------------------------------
class Fbank(nn.Module):
    def __init__(self, deltas=False, context=False, requires_grad=False, sample_rate=16000, f_min=0, f_max=None,
                 n_fft=400, n_mels=40, filter_shape="triangular", param_change_factor=1.0, param_rand_factor=0.0,
                 left_frames=5, right_frames=5, win_length=25, hop_length=10):
        super().__init__()
        if deltas or context or requires_grad or filter_shape != "triangular" or param_rand_factor != 0.0:
            raise NotImplementedError("ts-asr_amd.Fbank implements the configuration the TS-ASR recipes use "
                                      "(frozen triangular filters, no deltas/context)")
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.f_min, self.f_max = f_min, sample_rate / 2 if f_max is None else f_max
        self.win_length = int(round(sample_rate / 1000.0 * win_length))
        self.hop_length = int(round(sample_rate / 1000.0 * hop_length))
        self.top_db, self.amin = 80.0, 1e-10
        # fixed tables: built once on the host (the reference rebuilds the filter matrix on every call)
        self.register_buffer("window", torch.hamming_window(self.win_length), persistent=False)
        self.register_buffer("fbank_matrix", _mel_matrix(n_mels, n_fft, sample_rate, self.f_min, self.f_max), persistent=False)

    def forward(self, wav):
        C.require_gpu(wav)
        return ops.fbank(wav.float(), self.window, self.fbank_matrix, self.n_fft, self.hop_length, self.win_length,
                         self.top_db, self.amin)


def _mel_matrix(n_mels, n_fft, sample_rate, f_min, f_max):
    mel_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    mel_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    hz = 700.0 * (10.0 ** (torch.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
    band, centre = (hz[1:] - hz[:-1])[:-1], hz[1:-1]
    freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    slope = (freqs[None, :] - centre[:, None]) / band[:, None]
    return torch.clamp(torch.minimum(slope + 1.0, 1.0 - slope), min=0.0).t().contiguous()  # [n_fft/2+1, n_mels]


class InputNormalization(nn.Module):
    def __init__(self, mean_norm=True, std_norm=True, norm_type="global", avg_factor=None, requires_grad=False,
                 update_until_epoch=3):
        super().__init__()
        if norm_type != "sentence" or not mean_norm or not std_norm:
            raise NotImplementedError("ts-asr_amd.InputNormalization implements norm_type='sentence' (all TS-ASR YAMLs)")
        self.norm_type, self.eps, self.update_until_epoch = norm_type, 1e-10, update_until_epoch

    def forward(self, x, lengths, spk_ids=None, epoch=0):
        C.require_gpu(x, lengths)
        return ops.sentence_norm(x, abs_lengths_round(lengths, x.shape[1]), self.eps)



# ------------------------------------------------------------------------------------------------------
# A17 / A18 the two augmenters of compute_forward (TRAIN stage, ``augment: True``; train_librispeechmix_scratch.py:82-94)
# ------------------------------------------------------------------------------------------------------
class SpecAugment(nn.Module):
    """speechbrain.lobes.augment.SpecAugment (SB/lobes/augment.py:32-201), same constructor. Differences: the draws come from a
    device-side counter generator (one tiny kernel) instead of torch's CPU/GPU generators - distribution-identical, no host
    synchronisation, capturable into the step's hipGraph - and the result is a new tensor (the reference warps in place)."""

    def __init__(self, time_warp=True, time_warp_window=5, time_warp_mode="bicubic", freq_mask=True, freq_mask_width=(0, 20),
                 n_freq_mask=2, time_mask=True, time_mask_width=(0, 100), n_time_mask=2, replace_with_zero=True):
        super().__init__()
        assert time_warp or freq_mask or time_mask, "at least one of time_warp, time_mask, or freq_mask should be applied"
        if time_warp and time_warp_mode != "bicubic":
            raise NotImplementedError("ts-asr_amd.SpecAugment implements time_warp_mode='bicubic' (the default and all TS-ASR YAMLs)")
        as_range = lambda w: (0, w) if isinstance(w, int) else tuple(w)  # noqa: E731
        self.apply_time_warp, self.time_warp_window, self.time_warp_mode = time_warp, time_warp_window, time_warp_mode
        self.freq_mask, self.freq_mask_width, self.n_freq_mask = freq_mask, as_range(freq_mask_width), n_freq_mask
        self.time_mask, self.time_mask_width, self.n_time_mask = time_mask, as_range(time_mask_width), n_time_mask
        self.replace_with_zero = replace_with_zero

    def _counts(self):
        return (self.n_freq_mask if self.freq_mask else 0), (self.n_time_mask if self.time_mask else 0)

    def draw(self, x):
        """Device int32 table of this call's random numbers (layout: include/tsasr_hip.h, tsasr_specaug_draw)."""
        B, T, Fq = x.shape[0], x.shape[-2], x.shape[-1]
        nf, nt = self._counts()
        return ops.spec_augment_draw(B, T, Fq, self.time_warp_window if self.apply_time_warp else 0, nf, self.freq_mask_width, nt,
                                     self.time_mask_width, x.device)

    def forward(self, x, params=None):
        C.require_gpu(x)
        shape = x.shape
        if x.dim() == 4:                      # [B,C,T,F]: masks are drawn per (batch x channel) row, as the reference's view does
            x = x.reshape(-1, shape[2], shape[3])
        nf, nt = self._counts()
        y = ops.spec_augment_apply(x, self.draw(x) if params is None else params, nf, nt, self.replace_with_zero)
        return y.view(shape)


class Resample(nn.Module):
    """speechbrain.processing.speech_augmentation.Resample (SB/processing/speech_augmentation.py:504-820): windowed-sinc
    polyphase resampling. The filter bank is built once on the host with the reference's formulas; the resampling itself is one
    HIP launch instead of one conv1d + conv_transpose1d + two pads per phase."""

    def __init__(self, orig_freq=16000, new_freq=16000, lowpass_filter_width=6):
        super().__init__()
        self.orig_freq, self.new_freq, self.lowpass_filter_width = int(orig_freq), int(new_freq), lowpass_filter_width
        base = math.gcd(self.orig_freq, self.new_freq)
        self.conv_stride, self.output_samples = self.orig_freq // base, self.new_freq // base
        self.conv_transpose_stride = self.output_samples
        self._bank = {}

    def _filters(self, device):
        """(weights [P,W] fp32, first [P] int32) on ``device``: Hann-windowed sinc, cutoff 0.99 x the lower Nyquist (:758-820)."""
        if device not in self._bank:
            cutoff = 0.99 * 0.5 * min(self.orig_freq, self.new_freq)
            half = self.lowpass_filter_width / (2.0 * cutoff)
            out_t = torch.arange(0.0, self.output_samples) / self.new_freq
            lo = torch.ceil((out_t - half) * self.orig_freq)
            hi = torch.floor((out_t + half) * self.orig_freq)
            taps = torch.arange(int((hi - lo + 1).max()))
            dt = (lo[:, None] + taps[None, :]) / self.orig_freq - out_t[:, None]
            w = torch.where(dt.abs() < half, 0.5 * (1 + torch.cos(2 * math.pi * cutoff / self.lowpass_filter_width * dt)), torch.zeros_like(dt))
            safe = torch.where(dt == 0, torch.ones_like(dt), dt)
            w = w * torch.where(dt == 0, torch.full_like(dt, 2 * cutoff), torch.sin(2 * math.pi * cutoff * safe) / (math.pi * safe))
            self.first_indices, self.weights = lo, w / self.orig_freq
            self._bank[device] = (self.weights.to(device).contiguous(), lo.to(torch.int32).to(device))
        return self._bank[device]

    def forward(self, waveforms):
        if self.orig_freq == self.new_freq:
            return waveforms
        C.require_gpu(waveforms)
        if waveforms.dim() == 2:
            w, f = self._filters(waveforms.device)
            return ops.resample(waveforms, w, f, self.orig_freq, self.new_freq)
        if waveforms.dim() == 3:              # [B,L,C]: channels are resampled independently
            B, L, Cn = waveforms.shape
            w, f = self._filters(waveforms.device)
            y = ops.resample(waveforms.transpose(1, 2).reshape(B * Cn, L), w, f, self.orig_freq, self.new_freq)
            return y.view(B, Cn, -1).transpose(1, 2)
        raise ValueError("Input must be 2 or 3 dimensions")


class SpeedPerturb(nn.Module):
    """speechbrain.processing.speech_augmentation.SpeedPerturb (:435-501): one of ``speeds`` (percent) is picked per batch with
    the same CPU draws as the reference (torch.rand(1), torch.randint(len(speeds), (1,))) and applied by ``Resample``."""

    def __init__(self, orig_freq, speeds=[90, 100, 110], perturb_prob=1.0):  # noqa: B006  (reference signature)
        super().__init__()
        self.orig_freq, self.speeds, self.perturb_prob = orig_freq, list(speeds), perturb_prob
        self.samp_index = 0
        self.resamplers = [Resample(orig_freq=orig_freq, new_freq=orig_freq * speed // 100) for speed in self.speeds]

    def draw(self):
        """The reference's two CPU draws for the NEXT forward (torch.rand(1) against perturb_prob, torch.randint over the speeds),
        taken ahead of it: the host picks the speed before a hipGraph replay and the Brain keys its graphs by the result - a draw made
        inside a captured forward would be frozen into every replay. Returns (skip, speed index)."""
        skip = bool(torch.rand(1) > self.perturb_prob)
        if not skip:
            self.samp_index = int(torch.randint(len(self.speeds), (1,))[0])
        self._pending = (skip, self.samp_index)
        return self._pending

    def forward(self, waveform):
        skip, idx = self._pending if getattr(self, "_pending", None) is not None else self.draw()
        self._pending = None
        if skip:
            return waveform.clone()
        return self.resamplers[idx](waveform)

# ------------------------------------------------------------------------------------------------------
# A3 convolutional front-end
# ------------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    """Gives a child the attribute name the reference's state_dict uses (``conv`` / ``norm``)."""

    def __init__(self, name, child):
        super().__init__()
        self.add_module(name, child)


_FUSED_CONVBLOCK = True   # (False: conv kernel + 2 LayerNorm kernels + dropout-add kernel - the path of shapes the fused kernel does not take)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, in_freq, kernel_size=3, stride=2, padding="same"):
        super().__init__()
        self.padding, self.stride, self.k = padding, stride, kernel_size
        out_freq = (in_freq + 2 * (kernel_size // 2) - kernel_size) // stride + 1
        self.convs = nn.Module()
        self.convs.add_module("conv_0", _Holder("conv", nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride)))
        self.convs.add_module("norm_0", _Holder("norm", nn.LayerNorm([out_freq, out_channels])))
        self.reduce_conv = nn.Module()
        self.reduce_conv.add_module("conv", _Holder("conv", nn.Conv2d(in_channels, out_channels, 1, stride=stride)))
        self.reduce_conv.add_module("norm", _Holder("norm", nn.LayerNorm([out_freq, out_channels])))
        self.out_freq = out_freq

    def forward(self, x, dropout, training):
        """x [B,T,F,C] (channels-last all the way: this IS the NHWC layout of a conv over (T,F))."""
        c, n = self.convs.conv_0.conv, self.convs.norm_0.norm
        rc, rn = self.reduce_conv.conv.conv, self.reduce_conv.norm.norm
        if _FUSED_CONVBLOCK and ops.frontend_block_supported(x, c.weight.shape[0]):   # the whole block in one kernel per direction
            return ops.frontend_block(x, c.weight, c.bias, rc.weight, rc.bias, n.weight, n.bias, rn.weight, rn.bias, self.padding,
                                      0.01, 1e-5, dropout, training)
        y, r = ops.frontend_convs(x, c.weight, c.bias, rc.weight, rc.bias, self.padding)
        y = ops.layer_norm(y, n.weight, n.bias, 1e-5, act_slope=0.01)               # LN over [F,C] + LeakyReLU, one pass
        r = ops.layer_norm(r, rn.weight, rn.bias, 1e-5)
        return ops.dropout_add(y, None, r, 1.0, dropout, training, outer_p=dropout)   # Dropout(r + Dropout(y)), one pass


class ConvolutionFrontEnd(nn.Module):
    def __init__(self, input_shape, num_blocks=3, num_layers_per_block=5, out_channels=(128, 256, 512),
                 kernel_sizes=(3, 3, 3), strides=(1, 2, 2), dilations=(1, 1, 1), residuals=(True, True, True),
                 conv_module=None, activation=nn.LeakyReLU, norm=None, dropout=0.1, conv_bias=True, padding="same",
                 conv_init=None):
        super().__init__()
        if num_layers_per_block != 1 or not all(residuals[:num_blocks]) or activation is not nn.LeakyReLU \
                or any(s != 2 for s in strides[:num_blocks]) or any(d != 1 for d in dilations[:num_blocks]):
            raise NotImplementedError("ts-asr_amd.ConvolutionFrontEnd implements the TS-ASR YAML configuration: "
                                      "1 layer per block, stride 2, residual blocks, LayerNorm, LeakyReLU")
        self.dropout = dropout
        freq, cin = input_shape[-1], 1
        for i in range(num_blocks):
            blk = ConvBlock(cin, out_channels[i], freq, kernel_sizes[i], strides[i], padding)
            self.add_module(f"convblock_{i}", blk)
            freq, cin = blk.out_freq, out_channels[i]
        self.num_blocks = num_blocks

    def forward(self, x):
        C.require_gpu(x)
        x = _cd(x)
        if x.ndim == 3:
            x = x.unsqueeze(-1)
        for i in range(self.num_blocks):
            x = getattr(self, f"convblock_{i}")(x, self.dropout, self.training)
        return x


# ------------------------------------------------------------------------------------------------------
# A6-A9 Conformer block
# ------------------------------------------------------------------------------------------------------
class RelPosEncXL(nn.Module):
    """Symmetric table of the reference (SB/nnet/attention.py:352: the sin term keeps its sign in the future half):
    pe[i] = PE_abs(|T-1-i|), i = 0..2T-2. Cached per (T, dtype): it depends on nothing else."""

    def __init__(self, emb_dim):
        super().__init__()
        self.emb_dim = emb_dim
        inv_freq = torch.exp(torch.arange(0, emb_dim, 2, dtype=torch.float32) * -(math.log(10000.0) / emb_dim))
        self.register_buffer("inv_freq", inv_freq)
        self._cache = {}

    def forward(self, x):
        T = x.size(1)
        key = (T, x.device)
        if key not in self._cache:
            with torch.no_grad():
                pos = (torch.arange(2 * T - 1, device=x.device, dtype=torch.float32) - (T - 1)).abs().unsqueeze(-1)
                ang = pos * self.inv_freq.to(x.device)
                pe = torch.empty(2 * T - 1, self.emb_dim, device=x.device)
                pe[:, 0::2], pe[:, 1::2] = torch.sin(ang), torch.cos(ang)
                self._cache[key] = pe.unsqueeze(0)
        return self._cache[key]


_POS_CD = {}


def _pos_cd(pos_embs):
    """pos_embs[0] in the compute dtype. The table is a long-lived cached tensor (RelPosEncXL): cast it once, not once per layer
    and step (18 small cast kernels per step otherwise)."""
    if pos_embs.dtype == _COMPUTE_DTYPE:
        return pos_embs[0]
    key = (pos_embs.data_ptr(), tuple(pos_embs.shape), _COMPUTE_DTYPE)
    hit = _POS_CD.get(key)
    if hit is None or hit[0] is not pos_embs:
        _POS_CD[key] = hit = (pos_embs, pos_embs[0].to(_COMPUTE_DTYPE))
    return hit[1]


class RelPosMHAXL(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, vbias=False, vdim=None, mask_pos_future=False):
        super().__init__()
        if vbias or (vdim is not None and vdim != embed_dim):
            raise NotImplementedError("vbias / vdim are not used by the TS-ASR recipes")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.mask_pos_future = mask_pos_future
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.linear_pos = nn.Linear(embed_dim, embed_dim, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(self.head_dim, num_heads))
        self.pos_bias_v = nn.Parameter(torch.empty(self.head_dim, num_heads))
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.scale = 1.0 / math.sqrt(embed_dim)  # 1/sqrt(embed_dim), NOT 1/sqrt(head_dim) (attention.py:452)

    def forward(self, query, key, value, pos_embs, key_padding_mask=None, attn_mask=None, return_attn_weights=True,
                key_lens=None, causal=None):
        """Self-attention only (query is key is value, as ConformerEncoderLayer calls it). ``key_lens`` (int32 [B],
        valid keys per utterance) / ``causal`` are the device-side equivalents of key_padding_mask / the look-ahead
        attn_mask; when the reference-style masks are given instead they are converted (prefix masks only)."""
        C.require_gpu(query)
        if not (query is key and key is value):
            raise NotImplementedError("RelPosMHAXL is used as self-attention on this path")
        B, T, D = query.shape
        if key_lens is None and key_padding_mask is not None:
            key_lens = (~key_padding_mask).sum(-1).to(torch.int32)
        if causal is None:
            causal = attn_mask is not None
        o, attn = self._context(_cd(query), pos_embs, key_lens, causal, return_attn_weights)
        out = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        return (out, attn) if return_attn_weights else out

    def project_pos(self, pos_embs):
        """(pk, deferrable): pk = linear_pos(pos_embs) [2T-1, D] - depends on the weights and the positional table only, so the encoder
        computes it for ALL its layers ahead of the first one (the main stream then does it while it would otherwise wait for the speaker
        branch at the injection, instead of once per layer on the critical path). deferrable: d(pk) has ONE reader - that weight's
        gradient - when pk is the HIP GEMM of a leaf weight over the gradient-free table (ops._RelPosAttnFn.backward)."""
        pos = _pos_cd(pos_embs)
        w = self.linear_pos.weight
        return ops.matmul_nt(pos, w), bool(ops._gemm_ok(pos, w) and w.is_leaf and not pos.requires_grad)

    def _context(self, x, pos_embs, key_lens, causal, need_weights, pk=None):
        qkv = ops.matmul_nt(x, self.in_proj_weight)                          # [B,T,H*3*Dh] per-head interleaved Q|K|V
        pk, deferrable = self.project_pos(pos_embs) if pk is None else pk
        return ops.relpos_attention(qkv, pk, self.pos_bias_u, self.pos_bias_v, key_lens, self.num_heads, self.scale, causal,
                                    self.dropout if self.training else 0.0, need_weights, dpk_deferrable=deferrable)

    def forward_add(self, x, res, pos_embs, key_lens=None, causal=False, need_weights=False):
        """res + out_proj(attention(x)) with the projection bias and the residual add in one epilogue pass."""
        o, attn = self._context(_cd(x), pos_embs, key_lens, causal, need_weights)
        return ops.dropout_add(ops.matmul_nt(o, self.out_proj.weight), self.out_proj.bias, res), attn


class PositionalwiseFeedForward(nn.Module):
    def __init__(self, d_ffn, input_shape=None, input_size=None, dropout=0.0, activation=nn.ReLU):
        super().__init__()
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if input_size is None:
            input_size = input_shape[-1]
        self.ffn = nn.Sequential(nn.Linear(input_size, d_ffn), activation(), nn.Dropout(dropout), nn.Linear(d_ffn, input_size))


def _act_slope(activation):
    a = activation() if isinstance(activation, type) else activation
    if isinstance(a, nn.LeakyReLU):
        return a.negative_slope
    if isinstance(a, nn.ReLU):
        return 0.0
    raise NotImplementedError("ts-asr_amd kernels fuse LeakyReLU/ReLU (all TS-ASR YAMLs pass torch.nn.LeakyReLU)")


class ConvolutionModule(nn.Module):
    def __init__(self, input_size, kernel_size=31, bias=True, activation=nn.LeakyReLU, dropout=0.0, causal=False, dilation=1):
        super().__init__()
        if dilation != 1 or not bias:
            raise NotImplementedError("dilation/bias-free conv module is not used by the TS-ASR recipes")
        self.causal, self.kernel_size, self.dropout = causal, kernel_size, dropout
        self.padding = (kernel_size - 1) if causal else (kernel_size - 1) // 2
        self.slope = _act_slope(activation)
        self.layer_norm = nn.LayerNorm(input_size)
        self.bottleneck = nn.Sequential(nn.Conv1d(input_size, 2 * input_size, kernel_size=1), nn.GLU(dim=1))
        self.conv = nn.Conv1d(input_size, input_size, kernel_size, padding=self.padding, groups=input_size)
        self.after_conv = nn.Sequential(nn.LayerNorm(input_size), activation(), nn.Linear(input_size, input_size), nn.Dropout(dropout))

    def forward(self, x, mask=None, valid_lens=None):
        """``valid_lens`` int32 [B] replaces the boolean pad mask ([B,T,1], True = padded) of the reference."""
        if valid_lens is None and mask is not None:
            valid_lens = (~mask.squeeze(-1)).sum(-1).to(torch.int32)
        return self.forward_add(x, None, valid_lens)

    def core(self, y, last=True):
        """Everything between the module's LayerNorm and its last bias/dropout: y = LN(x) -> [.., D] (bias of after_conv[2] not added);
        last=False stops in front of the last point-wise convolution (the caller fuses it with what follows)."""
        y = ops.matmul_nt(y, self.bottleneck[0].weight)     # 1x1 conv D->2D: the [2D, D, 1] Parameter itself (bias below)
        y = ops.convmod_core(y, self.bottleneck[0].bias, self.conv.weight, self.conv.bias, self.after_conv[0].weight,
                             self.after_conv[0].bias, self.causal, 1e-5, self.slope)                # bias+GLU+depthwise+LN+act
        return ops.matmul_nt(y, self.after_conv[2].weight) if last else y

    def forward_add(self, x, res, valid_lens=None):
        """res + module(x) with dropout, pad masking and the residual add fused into the last pass."""
        y = self.core(ops.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias, 1e-5))
        return ops.dropout_add(y, self.after_conv[2].bias, res, 1.0, self.dropout, self.training, valid_lens)


class ConformerEncoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, nhead, kernel_size=31, kdim=None, vdim=None, activation=nn.LeakyReLU, bias=True,
                 dropout=0.0, causal=False, attention_type="RelPosMHAXL", chunk_size=0):
        super().__init__()
        # chunk_size (BUILD EXTENSION, not in the reference: BASELINE.json configs[4] "chunk=40 frames"): with causal=True and chunk_size > 1
        # the look-ahead mask becomes block-causal - a frame attends its whole chunk and everything before it; the convolution module and
        # the front-end stay strictly causal. 0 / 1 = the reference's look-ahead mask.
        self.chunk_size = int(chunk_size or 0)
        if attention_type != "RelPosMHAXL":
            raise NotImplementedError("the TS-ASR encoder always uses RelPosMHAXL (models/conformer.py:131-132)")
        self.mha_layer = RelPosMHAXL(num_heads=nhead, embed_dim=d_model, dropout=dropout, mask_pos_future=causal)
        self.convolution_module = ConvolutionModule(d_model, kernel_size, bias, activation, dropout, causal=causal)
        self.ffn_module1 = nn.Sequential(nn.LayerNorm(d_model), PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout, activation=activation), nn.Dropout(dropout))
        self.ffn_module2 = nn.Sequential(nn.LayerNorm(d_model), PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout, activation=activation), nn.Dropout(dropout))
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.drop = nn.Dropout(dropout)
        self.causal, self.dropout, self.slope = causal, dropout, _act_slope(activation)

    def _ffn_add(self, x, mod):
        """x + 0.5 * Dropout(PFF(LN(x)))  - macaron half-step (Conformer.py:243,258)."""
        ln, pff = mod[0], mod[1].ffn
        y = ops.layer_norm(x, ln.weight, ln.bias, 1e-5)
        y = ops.ffn_core(y, pff[0].weight, pff[0].bias, pff[3].weight, self.slope, self.dropout, self.training)   # two GEMMs, fused epilogues
        return ops.dropout_add(y, pff[3].bias, x, 0.5, self.dropout, self.training)

    def forward(self, x, src_mask=None, src_key_padding_mask=None, pos_embs=None, valid_lens=None, need_attn=True, prenorm=None,
                next_ln=None, pk=None):
        """Returns (x, attention weights [B,H,T,T] or None). The reference always materialises the weights
        (Conformer.py:247-254); the encoder passes need_attn=False unless return_attn is requested.
        Every ``residual + branch`` of Conformer.py:243-259 is fused with the LayerNorm that reads it (ops.add_layer_norm).
        The seam between two LAYERS is two LayerNorms in a row (this layer's norm2, then the next layer's first macaron LayerNorm or the
        encoder's final norm): the encoder may pass that next ``nn.LayerNorm`` as ``next_ln`` - a third value, next_ln(x), is then
        returned (same launch as norm2, same bits) - and hand it to the next layer as ``prenorm``."""
        if valid_lens is None and src_key_padding_mask is not None:
            valid_lens = (~src_key_padding_mask).sum(-1).to(torch.int32)
        tr, p, conv, mha = self.training, self.dropout, self.convolution_module, self.mha_layer
        ln1, pff1 = self.ffn_module1[0], self.ffn_module1[1].ffn
        ln2, pff2 = self.ffn_module2[0], self.ffn_module2[1].ffn
        x = _cd(x)
        if prenorm is None:
            y, x = ops.layer_norm_res(x, ln1.weight, ln1.bias, 1e-5)     # x is read twice (here and as the residual): one backward kernel sums both gradients
        else:
            y = prenorm                                                  # ln1(x), computed by the previous layer's last launch
        h = ops.ffn_core(y, pff1[0].weight, pff1[0].bias, pff1[3].weight, self.slope, p, tr)
        x, y = ops.add_layer_norm(h, pff1[3].bias, x, self.norm1.norm, 0.5, p, tr)                  # x + .5*drop(ffn1) ; norm1
        causal = self.causal or src_mask is not None
        o, attn = mha._context(y, pos_embs, valid_lens, (max(self.chunk_size, 1) if causal else 0), need_attn, pk=pk)   # pk: mha.project_pos(pos_embs), made ahead by the encoder
        # the two seams whose GEMM has K = d_model: projection, residual tail and LayerNorm in one launch (csrc/linear_ln.hip)
        x, y = ops.linear_add_layer_norm(o, mha.out_proj.weight, mha.out_proj.bias, x, conv.layer_norm)               # + skip ; conv LN
        c = conv.core(y, last=False)
        x, y = ops.linear_add_layer_norm(c, conv.after_conv[2].weight, conv.after_conv[2].bias, x, ln2, 1.0, conv.dropout, tr, valid_lens)   # + conv ; ffn2 LN
        h = ops.ffn_core(y, pff2[0].weight, pff2[0].bias, pff2[3].weight, self.slope, p, tr)
        if next_ln is not None and ops.add_layer_norm2_supported(h):
            x, z = ops.add_layer_norm2(h, pff2[3].bias, x, self.norm2.norm, next_ln, 0.5, p, tr, eps2=next_ln.eps)
            return x, attn, z
        _, x = ops.add_layer_norm(h, pff2[3].bias, x, self.norm2.norm, 0.5, p, tr)                  # norm2(x + .5*drop(ffn2))
        return (x, attn) if next_ln is None else (x, attn, None)
