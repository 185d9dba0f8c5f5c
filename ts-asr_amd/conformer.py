"""ConformerEncoder with speaker-embedding injection: host mirror of /root/reference models/conformer.py:31-288.

Same constructor arguments, forward signature, state_dict keys (``custom_src_module.layers.0.w.*``, ``layers.N.*``,
``norm.norm.*``, ``cat_proj.w.*`` / ``speaker_attn.att.*``, buffer ``positional_encoding.inv_freq``) and the same
xavier_normal_ re-initialisation of every >1-D parameter (models/conformer.py:284-287).
Differences underneath: padding is carried as int32 valid lengths on the device (no ``length_to_mask(..).max().item()``
host sync, SB/dataio/dataio.py:791), the ``cat`` injection never builds the [B,T,2D] concatenation
(cat_proj(cat[src, spk]) == src.W1^T + (spk.W2^T + b) broadcast over time).
"""
from typing import List, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn

from . import _capi as C
from . import ops
from .nnet import ConformerEncoderLayer, LayerNorm, Linear, RelPosEncXL, _cd, abs_lengths_round

__all__ = ["ConformerEncoder"]


class _SrcModule(nn.Module):
    """state_dict-compatible stand-in for speechbrain's ModuleList(Linear, Dropout) (keys ``layers.0.w.*``)."""

    def __init__(self, input_size, d_model, dropout):
        super().__init__()
        self.layers = nn.ModuleList([Linear(input_size=input_size, n_neurons=d_model, bias=True, combine_dims=False), nn.Dropout(dropout)])

    def forward(self, x):
        lin, drop = self.layers[0], self.layers[1]
        if x.ndim == 4 and lin.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        return ops.linear(_cd(x), lin.w.weight, lin.w.bias, None, drop.p, self.training)   # bias + Dropout in ONE epilogue pass


class _SpeakerAttention(nn.Module):
    """Holder for ``speaker_attn.att.*`` (torch.nn.MultiheadAttention parameters) - cross_attention injection."""

    def __init__(self, nhead, d_model, dropout, bias):
        super().__init__()
        self.att = nn.MultiheadAttention(embed_dim=d_model, num_heads=nhead, dropout=dropout, bias=bias)
        self.nhead, self.dropout = nhead, dropout

    def forward(self, src, spk, key_lens):
        B, T, D = src.shape
        S, H = spk.shape[1], self.nhead
        w, b = self.att.in_proj_weight, self.att.in_proj_bias
        q = ops.linear(src, w[:D], b[:D]).view(B, T, H, D // H).transpose(1, 2)
        kv = ops.linear(_cd(spk), w[D:], b[D:]).view(B, S, 2, H, D // H)
        if D // H <= 64:    # scores, softmax, dropout, .V in one HIP kernel each way (csrc/attention_f32.hip, exact fp32 arithmetic)
            o = ops.attention_f32(q.transpose(1, 2), kv[:, :, 0], kv[:, :, 1], H, 1.0 / (D // H) ** 0.5, key_lens, False,
                                  self.dropout if self.training else 0.0)     # [B,T,H,Dh] / [B,S,H,Dh] strided views
            return ops.linear(o, self.att.out_proj.weight, self.att.out_proj.bias)
        ops.lib_fallback("cross_attention injection", f"head dim {D // H} > 64")
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)).float() / (D // H) ** 0.5
        if key_lens is not None:
            s = s.masked_fill((torch.arange(S, device=src.device)[None, :] >= key_lens[:, None]).view(B, 1, 1, S), float("-inf"))
        p = F.dropout(torch.softmax(s, -1), self.dropout, self.training).to(v.dtype)
        o = torch.matmul(p, v).transpose(1, 2).reshape(B, T, D)
        return ops.linear(o, self.att.out_proj.weight, self.att.out_proj.bias)


class ConformerEncoder(nn.Module):
    def __init__(self, input_size, d_model=512, nhead=8, num_layers=6, d_ffn=2048, dropout=0.0, activation=nn.ReLU,
                 positional_encoding="fixed_abs_sine", kernel_size=31, bias=True, attention_type="RelPosMHAXL",
                 max_length=2500, causal=False, injection_mode: "Optional[str]" = "prod",
                 injection_after: "Union[int, List[int]]" = 0, chunk_size: int = 0):   # chunk_size: build extension (nnet.ConformerEncoderLayer)
        super().__init__()
        if attention_type != "RelPosMHAXL":
            raise NotImplementedError("attention_type must be RelPosMHAXL on this path")
        self.input_size, self.d_model, self.nhead, self.num_layers = input_size, d_model, nhead, num_layers
        self.d_ffn, self.dropout, self.kernel_size, self.causal = d_ffn, dropout, kernel_size, causal
        self.injection_mode = injection_mode
        self.injection_after = list(injection_after) if isinstance(injection_after, (list, tuple)) else [injection_after]
        self.positional_encoding = RelPosEncXL(d_model)
        self.custom_src_module = _SrcModule(input_size, d_model, dropout)
        self.layers = nn.ModuleList([
            ConformerEncoderLayer(d_ffn=d_ffn, nhead=nhead, d_model=d_model, dropout=dropout, activation=activation,
                                  kernel_size=kernel_size, bias=bias, causal=causal, attention_type=attention_type, chunk_size=chunk_size)
            for _ in range(num_layers)])
        self.norm = LayerNorm(d_model, eps=1e-6)
        if injection_mode == "cat":
            self.cat_proj = Linear(input_size=2 * d_model, n_neurons=d_model, bias=True)
        elif injection_mode == "cross_attention":
            self.speaker_attn = _SpeakerAttention(nhead, d_model, dropout, bias)
        elif injection_mode not in ("prod", "sum", None):
            raise NotImplementedError(injection_mode)
        for p in self.parameters():  # models/conformer.py:284-287
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    def forward(self, src, wav_len=None, speaker_embs=None, speaker_embs_length=None, return_attn=False):
        state = self.forward_pre(src, wav_len, speaker_embs, speaker_embs_length, return_attn)
        return self.forward_post(state, speaker_embs, speaker_embs_length)

    def forward_pre(self, src, wav_len=None, speaker_embs=None, speaker_embs_length=None, return_attn=False):
        """The part of forward() that does not need the speaker embedding: everything up to (not including) the first injection -
        the recipe runs it beside the speaker branch. Returns the state forward_post() continues from (``state["x"]`` may be replaced
        by a detached leaf: the captured step cuts the autograd graph there)."""
        C.require_gpu(src)
        if src.ndim == 4:
            b, t, c1, c2 = src.shape
            src = src.reshape(b, t, c1 * c2)
        T = src.shape[1]
        valid = abs_lengths_round(wav_len, T) if wav_len is not None else None
        x = self.custom_src_module(_cd(src))
        state = {"valid": valid, "attns": [], "return_attn": return_attn, "next": 0, "pre": None, "inject_first": False}
        if -1 in self.injection_after and speaker_embs is not None:
            state.update(x=x, pos=None, inject_first=True)
            return state
        pos = self.positional_encoding(x)
        state.update(x=x, pos=pos, pks=self._project_pos(pos))
        return self._run_layers(state, speaker_embs, speaker_embs_length, stop_at_injection=True)

    def _project_pos(self, pos):
        """Every layer's projected positional table, made ahead of the first layer (RelPosMHAXL.project_pos)."""
        from .nnet import _pos_cd
        pks = ops.project_many(_pos_cd(pos), [layer.mha_layer.linear_pos.weight for layer in self.layers])   # one launch for all layers
        if pks is not None:
            return [(pk, True) for pk in pks]     # (deferrable: RelPosMHAXL.project_pos - leaf weights, gradient-free table, HIP GEMM)
        return [layer.mha_layer.project_pos(pos) for layer in self.layers]

    def forward_post(self, state, speaker_embs=None, speaker_embs_length=None):
        if state["inject_first"]:
            speaker_embs = speaker_embs() if callable(speaker_embs) else speaker_embs
            state["x"] = self._inject_speaker_emb(state["x"], speaker_embs, speaker_embs_length)
            state["pos"] = self.positional_encoding(state["x"])
            state["pks"] = self._project_pos(state["pos"])
            state["inject_first"] = False
        state = self._run_layers(state, speaker_embs, speaker_embs_length, stop_at_injection=False)
        x = self.norm(state["x"]) if state["pre"] is None else state["pre"]
        return (x, state["attns"]) if state["return_attn"] else x

    def _run_layers(self, state, speaker_embs, speaker_embs_length, stop_at_injection):
        x, pos, valid, pre, n = state["x"], state["pos"], state["valid"], state["pre"], len(self.layers)
        pks = state.get("pks") or [None] * n
        return_attn = state["return_attn"]
        i = state["next"]
        if state.get("pending_injection"):   # forward_pre stopped right in front of this injection
            speaker_embs = speaker_embs() if callable(speaker_embs) else speaker_embs
            x = self._inject_speaker_emb(x, speaker_embs, speaker_embs_length)
            state["pending_injection"] = False
        while i < n:
            layer = self.layers[i]
            inject = i in self.injection_after and speaker_embs is not None
            # a layer's norm2 is followed by another LayerNorm of the same rows - the next layer's first macaron LayerNorm, or the final
            # norm (models/conformer.py:223-233) - unless the speaker embedding is injected in between: one launch for the pair
            nxt = None if inject else (self.norm.norm if i == n - 1 else self.layers[i + 1].ffn_module1[0])
            x, attn, pre = layer(x, pos_embs=pos, valid_lens=valid, need_attn=return_attn, prenorm=pre, next_ln=nxt, pk=pks[i]) if nxt is not None \
                else layer(x, pos_embs=pos, valid_lens=valid, need_attn=return_attn, prenorm=pre, pk=pks[i]) + (None,)
            if return_attn:
                state["attns"].append(attn.detach())
            i += 1
            if inject:
                if stop_at_injection:
                    state.update(x=x, pre=pre, next=i, pending_injection=True)
                    return state
                # a callable = "not needed before this point": the recipe computes the speaker branch on a second HIP stream and
                # joins it here, so it overlaps the mixture's front-end and the layers before the injection
                speaker_embs = speaker_embs() if callable(speaker_embs) else speaker_embs
                x = self._inject_speaker_emb(x, speaker_embs, speaker_embs_length)
        state.update(x=x, pre=pre, next=i)
        return state

    def _inject_speaker_emb(self, src, spk, spk_len):
        spk = _cd(spk)
        if self.injection_mode in ("prod", "sum"):
            if ops.inject_ok(src, spk):
                return ops.inject(src, spk, self.injection_mode)     # one HIP launch each way
            return src * spk if self.injection_mode == "prod" else src + spk   # (shapes the kernel does not take: D % 8 != 0)
        if self.injection_mode == "cat":
            D = self.d_model
            w, b = self.cat_proj.w.weight, self.cat_proj.w.bias
            if ops.linear_cols_ok(src, w, 0, D) and ops.linear_cols_ok(spk, w, D, D):   # the two column halves of the ONE weight, in place
                a, s = ops.linear_cols(src, w, None, 0, D), ops.linear_cols(spk, w, b, D, D)    # [B,T,D], [B,1,D]
                return ops.inject(a, s, "sum") if ops.inject_ok(a, s) else a + s                # (the broadcast add and its sum over time: HIP)
            return ops.linear(src, w[:, :D], None) + ops.linear(spk, w[:, D:], b)  # [B,T,D] + [B,1,D]
        if self.injection_mode == "cross_attention":
            klen = abs_lengths_round(spk_len, spk.shape[-2]) if spk_len is not None else None
            return self.speaker_attn(src, spk, klen)
        if self.injection_mode is None:
            return src
        raise NotImplementedError
