"""Operator layer between the host modules (nnet.py / conformer.py) and the device.

Two kinds of operator live here, and DESIGN.md lists which is which:
  * HIP  - hand-written gfx950 kernels behind the C-ABI (autograd.Functions in this file / rnnt.py);
  * GLUE - plain library calls through PyTorch-ROCm (hipBLASLt GEMMs, MIOpen conv/LSTM, rocFFT) that have no
           hand-written kernel yet. They run on the GPU, never on the CPU, and are the list of work for the
           next rounds. ``STATUS`` below is the machine-readable version of that list.
"""
import math

import torch
import torch.nn.functional as F

from . import _capi as C

STATUS = {
    "joint_logits": "HIP", "rnnt_loss": "HIP",
    "linear": "GLUE(hipBLASLt)", "lstm": "GLUE(MIOpen)", "frontend_conv": "GLUE(MIOpen)", "fbank.stft": "GLUE(rocFFT)",
    "layer_norm": "GLUE", "layer_norm2": "GLUE", "sentence_norm": "GLUE", "relpos_attention": "GLUE",
    "glu_dwconv_ln_act": "GLUE", "mask_time": "GLUE",
}


def _w(p, like):
    """Parameter in the activation dtype (fp32 master weights stay untouched; autograd casts the gradient back)."""
    return p if p is None or p.dtype == like.dtype else p.to(like.dtype)


# ---------------------------------------------------------------------------------------------------------
def linear(x, weight, bias=None, act_slope=None):
    y = F.linear(x, _w(weight, x), _w(bias, x))
    if act_slope is not None:
        y = F.leaky_relu(y, act_slope)
    return y


def lstm(x, rnn, hx=None):
    # MIOpen's LSTM runs in fp32 here (tiny: 28 -> 512, 121 steps); output returned in the activation dtype
    out, hn = rnn(x.float(), hx) if hx is not None else rnn(x.float())
    return out.to(x.dtype), hn


def layer_norm(x, weight, bias, eps):
    return F.layer_norm(x.float(), weight.shape, weight, bias, eps).to(x.dtype)


def layer_norm2(x, weight, bias, eps, act_slope=None):
    """LayerNorm over the last TWO dims ([F, C] of the front-end), optional fused LeakyReLU."""
    y = F.layer_norm(x.float(), weight.shape, weight, bias, eps)
    if act_slope is not None:
        y = F.leaky_relu(y, act_slope)
    return y.to(x.dtype)


def mask_time(x, valid_lens):
    """Zero frames t >= valid_lens[b] of x [B,T,D] (ConvolutionModule's masked_fill_, Conformer.py:113-114)."""
    T = x.shape[1]
    keep = torch.arange(T, device=x.device)[None, :] < valid_lens[:, None]
    return x * keep.unsqueeze(-1).to(x.dtype)


# ---------------------------------------------------------------------------------------------------------
def fbank(wav, window, fbank_matrix, n_fft, hop, win, top_db, amin):
    st = torch.stft(wav, n_fft, hop, win, window.to(wav.device), center=True, pad_mode="constant", normalized=False,
                    onesided=True, return_complex=True)
    power = (st.real ** 2 + st.imag ** 2).transpose(1, 2)
    mel = power @ fbank_matrix.to(wav.device)
    x_db = 10.0 * torch.log10(torch.clamp(mel, min=amin))
    return torch.maximum(x_db, x_db.amax(dim=(-2, -1), keepdim=True) - top_db)


def sentence_norm(x, abs_lens, eps):
    """Per-utterance mean / unbiased std over the first abs_lens[b] frames, applied to the whole padded row."""
    T = x.shape[1]
    m = (torch.arange(T, device=x.device)[None, :] < abs_lens[:, None]).unsqueeze(-1).to(x.dtype)
    n = abs_lens.to(x.dtype).view(-1, 1, 1)
    mean = (x * m).sum(1, keepdim=True) / n
    var = (((x - mean) * m) ** 2).sum(1, keepdim=True) / (n - 1)
    return (x - mean) / torch.clamp(var.sqrt(), min=eps)


def frontend_conv(x, weight, bias, k, stride, padding):
    """x [B,T,F,Cin] channels-last -> [B,T',F',Cout]. The reference convolves [B,C,F,T] (SB/nnet/CNN.py:629-676):
    its kernel axes are (F, T); here the tensor is viewed [B,C,T,F] so the kernel is transposed instead."""
    xin = x.permute(0, 3, 1, 2)  # [B,C,T,F] view of NHWC memory
    if k > 1:
        if padding == "same":
            xin = F.pad(xin, (k // 2, k // 2, k // 2, k // 2), mode="reflect")
        elif padding == "causal":
            xin = F.pad(xin, (k // 2, k // 2, k - 1, 0))
        else:
            raise ValueError("Padding must be 'same' or 'causal'. Got " + str(padding))
    w = _w(weight, x).transpose(2, 3)
    y = F.conv2d(xin, w, _w(bias, x), stride=stride)
    return y.permute(0, 2, 3, 1)  # [B,T',F',Cout]


# ---------------------------------------------------------------------------------------------------------
def relpos_attention(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, dropout_p, need_weights):
    """qkv [B,T,H*3*Dh] (per head Q|K|V interleaved), pk [2T-1, D]. Returns (context [B,T,D], weights or None).
    score[i,j] = ((q_i+u).k_j + (q_i+v).p_{j-i+T-1}) * scale ; -inf on j >= key_lens[b] and (causal) j > i."""
    B, T, _ = qkv.shape
    D = pk.shape[-1]
    Dh = D // H
    q, k, v = qkv.view(B, T, H, 3 * Dh).chunk(3, dim=-1)
    u = _w(pos_bias_u, qkv).reshape(-1).view(1, 1, H, Dh)
    vb = _w(pos_bias_v, qkv).reshape(-1).view(1, 1, H, Dh)
    p = pk.view(1, -1, H, Dh)
    ac = torch.matmul((q + u).transpose(1, 2), k.permute(0, 2, 3, 1))
    bd_raw = torch.matmul((q + vb).transpose(1, 2), p.permute(0, 2, 3, 1))
    idx = torch.arange(T, device=qkv.device)
    rel = (idx[None, :] - idx[:, None] + T - 1).expand(B, H, T, T)
    score = (ac + torch.gather(bd_raw, 3, rel)).float() * scale
    if causal:
        score = score.masked_fill(idx[None, :] > idx[:, None], float("-inf"))
    if key_lens is not None:
        score = score.masked_fill((idx[None, :] >= key_lens[:, None]).view(B, 1, 1, T), float("-inf"))
    attn = torch.softmax(score, dim=-1)
    pa = F.dropout(attn, dropout_p, True) if dropout_p > 0 else attn
    o = torch.matmul(pa.to(v.dtype), v.transpose(1, 2)).transpose(1, 2).reshape(B, T, D)
    return o, (attn if need_weights else None)


def glu_dwconv_ln_act(y2, conv_w, conv_b, ln_w, ln_b, causal, eps, slope):
    """y2 [B,T,2D] -> GLU -> depthwise conv over time (K taps, 'same' or causal left pad) -> LayerNorm(D) -> LeakyReLU."""
    D = y2.shape[-1] // 2
    K = conv_w.shape[-1]
    g = y2[..., :D] * torch.sigmoid(y2[..., D:])
    gt = g.transpose(1, 2)
    gt = F.pad(gt, (K - 1, 0)) if causal else F.pad(gt, ((K - 1) // 2, (K - 1) // 2))
    c = F.conv1d(gt, _w(conv_w, y2), _w(conv_b, y2), groups=D).transpose(1, 2)
    c = F.layer_norm(c.float(), (D,), ln_w, ln_b, eps)
    return F.leaky_relu(c, slope).to(y2.dtype)
