"""CPU ORACLE for the TS-ASR Conformer-Transducer hot path  (TEST INFRASTRUCTURE).

A plain restatement (torch CPU ops, fp32, functional style: every function takes the tensors
and a ``{state_dict key: tensor}`` mapping) of what the reference computes on this path.
It is the checker for the HIP path and the ``cpu_baseline`` leg of bench.py. It is never
imported by the product package ``ts-asr_amd`` (which fails loudly without its HIP library).

Pinned against the reference itself: ``tests/golden/*.npz`` were produced by importing the
reference in the build container (``oracle/gen_golden.py``); ``tests/test_oracle_golden.py``
checks every function below against them. The RNN-T loss lives in a third-party dependency the
reference does not vendor (torchaudio.functional.rnnt_loss); it is restated in
``oracle/rnnt_ref.c`` / ``oracle/rnnt_ref.py`` and pinned by the reference's own known-answer
test (vendor/speechbrain/tests/unittests/test_losses.py:109-152).

All ``file:line`` citations are relative to /root/reference; SB = vendor/speechbrain/speechbrain.
"""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.01  # torch.nn.LeakyReLU default, hparams conformer-t_scratch.yaml:166


# ----------------------------------------------------------------------------------------------
# A1  Fbank  (SB/lobes/features.py:130-147, SB/processing/features.py:102-178,317-348,482-552,683-704)
# ----------------------------------------------------------------------------------------------
def mel_filterbank(n_mels=80, n_fft=512, sample_rate=16000, f_min=0.0, f_max=None):
    """[n_fft//2+1, n_mels] triangular filters, SB/processing/features.py:463-477,578-602."""
    f_max = sample_rate / 2 if f_max is None else f_max
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)  # noqa: E731  (:555-566)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)  # (:568-576)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    all_freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    slope = (all_freqs[None, :] - f_central[:, None]) / band[:, None]
    fb = torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0)  # max(0, min(left, right))
    return fb.t().contiguous()


def fbank(wav, n_fft=512, n_mels=80, win_ms=32, hop_ms=10, sample_rate=16000, top_db=80.0, amin=1e-10):
    """wav [B,L] -> log-mel [B, 1+L//hop, n_mels]."""
    win = int(round(sample_rate / 1000.0 * win_ms))
    hop = int(round(sample_rate / 1000.0 * hop_ms))
    window = torch.hamming_window(win)  # periodic (torch default), SB features.py:132
    st = torch.stft(wav, n_fft, hop, win, window, center=True, pad_mode="constant",
                    normalized=False, onesided=True, return_complex=True)
    power = (st.real ** 2 + st.imag ** 2).transpose(1, 2)  # spectral_magnitude power=1 (:339-345)
    mel = power @ mel_filterbank(n_mels, n_fft, sample_rate)
    x_db = 10.0 * torch.log10(torch.clamp(mel, min=amin))  # multiplier 10, ref_value 1 -> db_multiplier 0
    floor = x_db.amax(dim=(-2, -1), keepdim=True) - top_db  # per utterance incl. padded frames (:700-703)
    return torch.maximum(x_db, floor)


# ----------------------------------------------------------------------------------------------
# A2  InputNormalization(norm_type="sentence")  (SB/processing/features.py:1012-1025,1107-1132)
# ----------------------------------------------------------------------------------------------
def sentence_norm(x, rel_lens, eps=1e-10):
    out = torch.empty_like(x)
    for b in range(x.shape[0]):
        n = int(torch.round(rel_lens[b] * x.shape[1]).int())
        mu = x[b, :n].mean(dim=0)
        sd = torch.clamp(x[b, :n].std(dim=0), min=eps)  # unbiased
        out[b] = (x[b] - mu) / sd  # applied to the whole padded row
    return out


# ----------------------------------------------------------------------------------------------
# A3  ConvolutionFrontEnd  (SB/lobes/models/convolution.py:103-266, SB/nnet/CNN.py:629-711)
# ----------------------------------------------------------------------------------------------
def _conv2d_sb(x_bcft, w, b, stride, padding):
    """x is [B,C,F,T] (the layout SB's Conv2d convolves in after transpose(1,-1))."""
    k = w.shape[-1]
    if k == 1:
        return F.conv2d(x_bcft, w, b, stride=stride)
    if padding == "same":  # reflect pad k//2 on T and F (CNN.py:678-711, 1488-1489)
        x_bcft = F.pad(x_bcft, (k // 2, k // 2, k // 2, k // 2), mode="reflect")
    elif padding == "causal":  # zero pad (k-1, 0) on T, (k//2, k//2) on F (CNN.py:649-657)
        x_bcft = F.pad(x_bcft, (k - 1, 0, k // 2, k // 2))
    else:
        raise ValueError(padding)
    return F.conv2d(x_bcft, w, b, stride=stride)


def conv_block(x, sd, p, padding):
    """x [B,T,F,C] -> [B,T/2,F/2,C']; ConvBlock.forward convolution.py:260-266 (dropout = identity)."""
    xt = x.transpose(1, -1)  # [B,C,F,T]
    y = _conv2d_sb(xt, sd[p + "convs.conv_0.conv.weight"], sd[p + "convs.conv_0.conv.bias"], 2, padding).transpose(1, -1)
    nw = sd[p + "convs.norm_0.norm.weight"]
    y = F.layer_norm(y, nw.shape, nw, sd[p + "convs.norm_0.norm.bias"], 1e-5)
    y = F.leaky_relu(y, LRELU_SLOPE)
    r = _conv2d_sb(xt, sd[p + "reduce_conv.conv.conv.weight"], sd[p + "reduce_conv.conv.conv.bias"], 2, "same").transpose(1, -1)
    rw = sd[p + "reduce_conv.norm.norm.weight"]
    r = F.layer_norm(r, rw.shape, rw, sd[p + "reduce_conv.norm.norm.bias"], 1e-5)
    return y + r


def frontend(feats, sd, padding="same"):
    """feats [B,T,80] -> [B,T/4,20,128]."""
    x = feats.unsqueeze(-1)  # C=1 (Conv2d unsqueeze path, CNN.py:641-642)
    x = conv_block(x, sd, "convblock_0.", padding)
    return conv_block(x, sd, "convblock_1.", padding)


# ----------------------------------------------------------------------------------------------
# A6  RelPosEncXL  (SB/nnet/attention.py:327-359)  - symmetric table, see SURVEY.md section 8a
# ----------------------------------------------------------------------------------------------
def relpos_table(T, D, dtype=torch.float32):
    inv_freq = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(-1)
    pe = torch.zeros(T, D)
    pe[:, 0::2] = torch.sin(pos * inv_freq)
    pe[:, 1::2] = torch.cos(pos * inv_freq)
    # past half = flip(pe), future half = pe[1:] with the SAME sin sign (attention.py:352)
    return torch.cat([torch.flip(pe, (0,)), pe[1:]], dim=0).unsqueeze(0).to(dtype)


# ----------------------------------------------------------------------------------------------
# A7  RelPosMHAXL  (SB/nnet/attention.py:485-639)
# ----------------------------------------------------------------------------------------------
def relpos_core(qkv, pk, pos_bias_u, pos_bias_v, H, scale, key_padding_mask=None, causal=False):
    """Attention proper of RelPosMHAXL, after the input / positional projections and before out_proj (SB/nnet/attention.py:549-633):
    qkv [B,T,3D] per-head interleaved Q|K|V, pk [2T-1, D], biases in their (Dh,H) storage. Returns (context [B,T,D], attn [B,H,T,T]).
    relpos_mha below is this function between the projections, so the golden vectors that pin relpos_mha pin it too; the GPU tests
    of the fused attention kernels call it directly."""
    B, T, D3 = qkv.shape
    D = D3 // 3
    Dh = D // H
    q, k, v = qkv.view(B, T, H, 3 * Dh).chunk(3, dim=-1)
    pk = pk.view(1, -1, H, Dh)  # [1,2T-1,H,Dh]
    u = pos_bias_u.view(1, 1, H, Dh)  # reinterpreting view of (Dh,H) storage (:586-592)
    vb = pos_bias_v.view(1, 1, H, Dh)
    ac = torch.matmul((q + u).transpose(1, 2), k.permute(0, 2, 3, 1))  # [B,H,T,T]
    bd_raw = torch.matmul((q + vb).transpose(1, 2), pk.permute(0, 2, 3, 1))  # [B,H,T,2T-1]
    # rel_shift closed form (:468-483): BD[i,j] = BDraw[i, j - i + T - 1]
    idx = (torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1)  # [T,T]
    bd = torch.gather(bd_raw, 3, idx.expand(B, H, T, T))
    score = (ac + bd) * scale
    if causal:   # float look-ahead mask is added (:615-616); causal = C > 1: the build's block-causal extension (chunks of C frames:
        ii = torch.arange(T)   # a frame sees its whole chunk and everything before it) - not a reference feature, see DESIGN.md
        lim = ii if int(causal) <= 1 else (ii // int(causal) + 1) * int(causal) - 1
        score = score + torch.zeros(T, T).masked_fill(ii[None, :] > lim[:, None], float("-inf"))
    if key_padding_mask is not None:
        score = score.masked_fill(key_padding_mask.view(B, 1, 1, T), float("-inf"))
    attn = torch.softmax(score, dim=-1)
    o = torch.matmul(attn, v.transpose(1, 2)).transpose(1, 2).reshape(B, T, D)
    return o, attn


def relpos_mha(x, pe, sd, p, H, key_padding_mask=None, causal=False, return_attn=False):
    B, T, D = x.shape
    qkv = x @ sd[p + "in_proj_weight"].t()  # per-head interleaved (:549-553)
    pk = pe @ sd[p + "linear_pos.weight"].t()
    pk = pk.reshape(-1, D)
    o, attn = relpos_core(qkv, pk, sd[p + "pos_bias_u"], sd[p + "pos_bias_v"], H, 1.0 / math.sqrt(D),  # 1/sqrt(embed_dim)  (:452,604)
                          key_padding_mask, causal)
    o = o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]
    return (o, attn) if return_attn else o


# ----------------------------------------------------------------------------------------------
# A8  ConvolutionModule  (SB/lobes/models/transformer/Conformer.py:73-115)
# ----------------------------------------------------------------------------------------------
def conv_module(x, sd, p, pad_mask=None, causal=False):
    D = x.shape[-1]
    K = sd[p + "conv.weight"].shape[-1]
    y = F.layer_norm(x, (D,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    y = y @ sd[p + "bottleneck.0.weight"].squeeze(-1).t() + sd[p + "bottleneck.0.bias"]  # 1x1 conv D->2D
    y = y[..., :D] * torch.sigmoid(y[..., D:])  # GLU over channels
    yt = y.transpose(1, 2)
    if causal:  # pad K-1 both sides then chomp the last K-1  == left pad K-1 (:68-71,108-110)
        yt = F.conv1d(F.pad(yt, (K - 1, 0)), sd[p + "conv.weight"], sd[p + "conv.bias"], groups=D)
    else:
        yt = F.conv1d(yt, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=(K - 1) // 2, groups=D)
    y = yt.transpose(1, 2)
    y = F.layer_norm(y, (D,), sd[p + "after_conv.0.weight"], sd[p + "after_conv.0.bias"], 1e-5)
    y = F.leaky_relu(y, LRELU_SLOPE)
    y = y @ sd[p + "after_conv.2.weight"].t() + sd[p + "after_conv.2.bias"]
    if pad_mask is not None:
        y = y.masked_fill(pad_mask.unsqueeze(-1), 0.0)
    return y


# ----------------------------------------------------------------------------------------------
# A9  ConformerEncoderLayer  (Conformer.py:194-260; PFF SB/nnet/attention.py:820-836)
# ----------------------------------------------------------------------------------------------
def ffn_module(x, sd, p):
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[p + "0.weight"], sd[p + "0.bias"], 1e-5)
    y = F.leaky_relu(y @ sd[p + "1.ffn.0.weight"].t() + sd[p + "1.ffn.0.bias"], LRELU_SLOPE)
    return y @ sd[p + "1.ffn.3.weight"].t() + sd[p + "1.ffn.3.bias"]


def conformer_layer(x, pe, sd, p, H, key_padding_mask=None, causal=False):
    D = x.shape[-1]
    x = x + 0.5 * ffn_module(x, sd, p + "ffn_module1.")
    y = F.layer_norm(x, (D,), sd[p + "norm1.norm.weight"], sd[p + "norm1.norm.bias"], 1e-5)
    x = relpos_mha(y, pe, sd, p + "mha_layer.", H, key_padding_mask, causal) + x
    x = x + conv_module(x, sd, p + "convolution_module.", key_padding_mask, causal)
    x = x + 0.5 * ffn_module(x, sd, p + "ffn_module2.")
    return F.layer_norm(x, (D,), sd[p + "norm2.norm.weight"], sd[p + "norm2.norm.bias"], 1e-5)


# ----------------------------------------------------------------------------------------------
# A4/A5  ConformerEncoder with speaker-embedding injection  (models/conformer.py:169-282)
# ----------------------------------------------------------------------------------------------
def length_to_mask(abs_len, max_len):
    """SB/dataio/dataio.py:758-803: True for valid positions."""
    return torch.arange(max_len)[None, :] < abs_len[:, None]


def cross_attention(q_in, kv, sd, p, H, key_padding_mask=None):
    """nn.MultiheadAttention restated (SB wrapper attention.py:642-775 -> torch MHA, batch-first view)."""
    B, T, D = q_in.shape
    S = kv.shape[1]
    Dh = D // H
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = (q_in @ w[:D].t() + b[:D]).view(B, T, H, Dh).transpose(1, 2)
    k = (kv @ w[D:2 * D].t() + b[D:2 * D]).view(B, S, H, Dh).transpose(1, 2)
    v = (kv @ w[2 * D:].t() + b[2 * D:]).view(B, S, H, Dh).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(Dh)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.view(B, 1, 1, S), float("-inf"))
    o = torch.matmul(torch.softmax(s, -1), v).transpose(1, 2).reshape(B, T, D)
    return o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def inject_speaker(src, spk, spk_lens, sd, p, mode, H):
    if mode == "prod":
        return src * spk
    if mode == "sum":
        return src + spk
    if mode == "cat":
        cat = torch.cat([src, spk.expand(-1, src.shape[1], -1)], dim=-1)
        return cat @ sd[p + "cat_proj.w.weight"].t() + sd[p + "cat_proj.w.bias"]
    if mode == "cross_attention":
        kpm = None
        if spk_lens is not None:
            kpm = ~length_to_mask((spk_lens * spk.shape[1]).round(), spk.shape[1])
        return cross_attention(src, spk, sd, p + "speaker_attn.att.", H, kpm)
    if mode is None:
        return src
    raise NotImplementedError(mode)


def conformer_encoder(src, rel_lens, sd, p, H, num_layers, spk=None, spk_lens=None,
                      injection_mode="cat", injection_after=(0,), causal=False):
    if src.ndim == 4:
        src = src.reshape(src.shape[0], src.shape[1], -1)
    T = src.shape[1]
    kpm = None
    if rel_lens is not None:
        kpm = ~length_to_mask((rel_lens * T).round(), T)  # models/conformer.py:270-275
    x = src @ sd[p + "custom_src_module.layers.0.w.weight"].t() + sd[p + "custom_src_module.layers.0.w.bias"]
    if -1 in injection_after and spk is not None:
        x = inject_speaker(x, spk, spk_lens, sd, p, injection_mode, H)
    pe = relpos_table(T, x.shape[-1])
    for i in range(num_layers):
        x = conformer_layer(x, pe, sd, f"{p}layers.{i}.", H, kpm, causal)
        if i in injection_after and spk is not None:
            x = inject_speaker(x, spk, spk_lens, sd, p, injection_mode, H)
    return F.layer_norm(x, (x.shape[-1],), sd[p + "norm.norm.weight"], sd[p + "norm.norm.bias"], 1e-6)


# ----------------------------------------------------------------------------------------------
# A10  speaker branch pooling  (train_librispeechmix_scratch.py:52-64)
# ----------------------------------------------------------------------------------------------
def masked_mean_pool(x, rel_lens):
    T = x.shape[1]
    n = (rel_lens * T).ceil().clamp(max=T)
    mask = length_to_mask(n, T).to(x.dtype)[..., None]
    return (x * mask).sum(dim=1, keepdim=True) / mask.sum(dim=1, keepdim=True)


# ----------------------------------------------------------------------------------------------
# A12  one-hot Embedding + LSTM predictor  (SB/nnet/embedding.py:70-114, SB/nnet/RNN.py:25-52,244-278)
# ----------------------------------------------------------------------------------------------
def one_hot_embedding(tokens, vocab_size, blank_id=0):
    """Row blank_id is zeros; other rows are one-hot of dimension vocab_size-1."""
    w = torch.zeros(vocab_size, vocab_size - 1)
    eye = torch.eye(vocab_size - 1)
    w[blank_id + 1:] = eye[blank_id:]
    if blank_id != 0:
        w[:blank_id] = eye[:blank_id]
    return w[tokens.long()]


def lstm(x, sd, p, rel_lens=None):
    """Single-layer batch-first LSTM, gate order i,f,g,o (torch.nn.LSTM). Packed-sequence semantics:
    outputs beyond each sequence's length are zero; output is trimmed to the longest length."""
    B, T, _ = x.shape
    wi, wh = sd[p + "rnn.weight_ih_l0"], sd[p + "rnn.weight_hh_l0"]
    bi, bh = sd[p + "rnn.bias_ih_l0"], sd[p + "rnn.bias_hh_l0"]
    Hd = wh.shape[1]
    if rel_lens is None:
        abs_len = torch.full((B,), T, dtype=torch.long)
    else:
        abs_len = (rel_lens * T).long()  # pack_padded_sequence truncates the float lengths (RNN.py:35)
    Tm = int(abs_len.max())
    h = x.new_zeros(B, Hd)
    c = x.new_zeros(B, Hd)
    xs = x @ wi.t() + bi + bh
    outs = []
    for t in range(Tm):
        g = xs[:, t] + h @ wh.t()
        i, f, gg, o = g.chunk(4, dim=-1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        live = (t < abs_len)[:, None]
        c = torch.where(live, c_new, c)
        h = torch.where(live, h_new, h)
        outs.append(torch.where(live, h_new, torch.zeros_like(h_new)))
    return torch.stack(outs, dim=1), (h, c)


# ----------------------------------------------------------------------------------------------
# A11/A13  projections, joint, head  (SB/nnet/linear.py:15-78, transducer_joint.py:73-95)
# ----------------------------------------------------------------------------------------------
def linear(x, sd, p):
    return x @ sd[p + "w.weight"].t() + sd[p + "w.bias"]


def joint_logits(enc_proj, dec_proj, sd, p_head):
    j = F.leaky_relu(enc_proj[:, :, None, :] + dec_proj[:, None, :, :], LRELU_SLOPE)
    return linear(j, sd, p_head)


# ----------------------------------------------------------------------------------------------
# A15  Noam schedule  (SB/nnet/schedulers.py:403-436)
# ----------------------------------------------------------------------------------------------
def noam_lr(lr0, n_steps, n_warmup):
    return lr0 * (n_warmup ** 0.5) * min(n_steps ** -0.5, n_steps * n_warmup ** -1.5)


# ----------------------------------------------------------------------------------------------
# whole forward as TSASR.compute_forward wires it (train_librispeechmix_scratch.py:34-148)
# ``sd`` holds every module's state_dict under its Brain module name ("encoder.layers.0...." etc.)
# ----------------------------------------------------------------------------------------------
def compute_forward(batch, sd, cfg, injection_mode="cat", causal=False, frontend_padding="same",
                    from_feats=False, collect=None):
    c = {} if collect is None else collect
    H = cfg["nhead"]
    if from_feats:  # bench workload: mel features given (already normalised), SURVEY.md section 8d
        sf, f = batch.get("enroll_feats"), batch["mixed_feats"]
    else:
        sf = sentence_norm(fbank(batch["enroll_sig"]), batch["enroll_lens"]) if "enroll_sig" in batch and "enroll_emb" not in batch else None
        f = sentence_norm(fbank(batch["mixed_sig"]), batch["mixed_lens"])
        c["spk_norm"], c["norm"] = sf, f
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}  # noqa: E731
    spk = None
    if "enroll_emb" in batch:   # pretrained-speaker variant (train_librispeechmix_pretrained.py:45-63,80): the frozen speaker encoder's
        spk = linear(batch["enroll_emb"], sd, "speaker_proj.")   # embedding [B,1,E] (or hidden states [B,S,E]) is an input; only speaker_proj is on the path
        c["spk_emb"] = spk
    elif "speaker_encoder.norm.norm.weight" in sd:
        se = frontend(sf, sub("speaker_frontend."), "same")
        se = conformer_encoder(se, batch["enroll_lens"], sd, "speaker_encoder.", H, cfg["speaker_num_layers"])
        c["spk_enc"] = se
        if injection_mode != "cross_attention":
            se = masked_mean_pool(se, batch["enroll_lens"])
        c["spk_pool"] = se
        spk = linear(se, sd, "speaker_proj.")
        c["spk_emb"] = spk
    f = frontend(f, sub("frontend."), frontend_padding)
    c["frontend"] = f
    e = conformer_encoder(f, batch["mixed_lens"], sd, "encoder.", H, cfg["encoder_num_layers"], spk,
                          batch.get("enroll_lens"), injection_mode, (0,), causal)   # no speaker branch (train_librispeechmix_none.py:78): spk is None
    c["enc"] = e
    e = linear(e, sd, "encoder_proj.")
    c["enc_proj"] = e
    emb = one_hot_embedding(batch["tokens_bos"], cfg["vocab_size"], cfg["blank_index"])
    d, _ = lstm(emb, sd, "decoder.", batch["tokens_bos_lens"])
    c["dec"] = d
    d = linear(d, sd, "decoder_proj.")
    c["dec_proj"] = d
    logits = joint_logits(e, d, sd, "transducer_head.")
    c["logits"] = logits
    return logits


# ----------------------------------------------------------------------------------------------
# f1  greedy transducer search  (SB/decoders/transducer.py:138-218): at most one symbol per frame
# ----------------------------------------------------------------------------------------------
def greedy_decode(enc_proj, sd, cfg):
    B, T, _ = enc_proj.shape
    V, blank = cfg["vocab_size"], cfg["blank_index"]
    wi, wh = sd["decoder.rnn.weight_ih_l0"], sd["decoder.rnn.weight_hh_l0"]
    bias = sd["decoder.rnn.bias_ih_l0"] + sd["decoder.rnn.bias_hh_l0"]
    Hd = wh.shape[1]

    def step(tok, h, c):
        x = one_hot_embedding(tok, V, blank)
        g = x @ wi.t() + bias + h @ wh.t()
        i, f, gg, o = g.chunk(4, dim=-1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return linear(h2, sd, "decoder_proj."), h2, c2

    tok = torch.full((B,), blank, dtype=torch.long)
    out_pn, h, c = step(tok, torch.zeros(B, Hd), torch.zeros(B, Hd))
    hyps = [[] for _ in range(B)]
    for t in range(T):
        lg = linear(F.leaky_relu(enc_proj[:, t] + out_pn, LRELU_SLOPE), sd, "transducer_head.")
        pos = torch.log_softmax(lg, -1).argmax(-1)
        upd = pos != blank
        if upd.any():
            for b in torch.nonzero(upd).flatten().tolist():
                hyps[b].append(int(pos[b]))
            o2, h2, c2 = step(torch.where(upd, pos, tok), h, c)
            m = upd[:, None]
            out_pn, h, c = torch.where(m, o2, out_pn), torch.where(m, h2, h), torch.where(m, c2, c)
    return hyps

# ----------------------------------------------------------------------------------------------
# f1  beam transducer search  (SB/decoders/transducer.py:220-373), per utterance, no LM:
#   A = hypotheses still to be extended at this frame, B = hypotheses that emitted blank at this frame (next frame's A).
#   Until |B| >= beam: take the best a in A by length-normalised score (logp / len(prediction), blank prefix included); stop when
#   the best b in B (same key) has logp >= state_beam + logp(a); run the predictor on a's last token; take the beam best symbols
#   of log_softmax(joint(enc[t], pn)); blank -> copy of a with the score added goes to B (state unchanged); a non-blank symbol
#   within expand_beam of the best non-blank -> a extended (new predictor state) goes back to A.
#   Result: best of B by the same key; reported score = logp / len(prediction).
# ----------------------------------------------------------------------------------------------
def beam_decode(enc_proj, sd, cfg, beam_size=4, state_beam=2.3, expand_beam=2.3):
    B, T, _ = enc_proj.shape
    V, blank = cfg["vocab_size"], cfg["blank_index"]
    wi, wh = sd["decoder.rnn.weight_ih_l0"], sd["decoder.rnn.weight_hh_l0"]
    bias = sd["decoder.rnn.bias_ih_l0"] + sd["decoder.rnn.bias_hh_l0"]
    Hd = wh.shape[1]

    def pn_step(tok, state):
        h, c = state if state is not None else (torch.zeros(1, Hd), torch.zeros(1, Hd))
        x = one_hot_embedding(torch.tensor([tok]), V, blank)
        g = x @ wi.t() + bias + h @ wh.t()
        i, f, gg, o = g.chunk(4, dim=-1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return linear(h2, sd, "decoder_proj."), (h2, c2)

    def key(hyp):
        return hyp[1] / len(hyp[0])

    hyps_out, scores_out = [], []
    for b in range(B):
        beam = [([blank], 0.0, None)]            # (prediction, logp, predictor state)
        for t in range(T):
            A, beam = beam, []
            while len(beam) < beam_size:
                a = max(A, key=key)
                if beam and max(beam, key=key)[1] >= state_beam + a[1]:
                    break
                A.remove(a)
                out_pn, new_state = pn_step(a[0][-1], a[2])
                lg = linear(F.leaky_relu(enc_proj[b, t][None] + out_pn, LRELU_SLOPE), sd, "transducer_head.")
                logp, pos = torch.topk(torch.log_softmax(lg, -1).view(-1), k=beam_size)
                best_nonblank = logp[0] if int(pos[0]) != blank else logp[1]
                for j in range(beam_size):
                    sc = a[1] + float(logp[j])
                    if int(pos[j]) == blank:
                        beam.append((a[0][:], sc, a[2]))
                    elif logp[j] >= best_nonblank - expand_beam:
                        A.append((a[0] + [int(pos[j])], sc, new_state))
        best = max(beam, key=key)
        hyps_out.append(best[0][1:])
        scores_out.append(best[1] / len(best[0]))
    return hyps_out, scores_out


# ----------------------------------------------------------------------------------------------
# A17  SpecAugment  (SB/lobes/augment.py:32-201; recipe settings conformer-t_scratch.yaml:132-142;
#      applied to the normalised features in TRAIN stage, train_librispeechmix_scratch.py:91-94)
# ----------------------------------------------------------------------------------------------
def _cubic_taps(t, A=-0.75):
    """Cubic-convolution weights of the four taps around a source position with fractional part ``t`` (what
    F.interpolate(mode="bicubic") uses; float32 throughout like its CPU kernel)."""
    import numpy as np
    t = t.astype(np.float32)
    A = np.float32(A)
    one, two, three, four, five, eight = (np.float32(v) for v in (1, 2, 3, 4, 5, 8))

    def near(x):   # |x| <= 1
        return ((A + two) * x - (A + three)) * x * x + one

    def far(x):    # 1 < |x| < 2
        return ((A * x - five * A) * x + eight * A) * x - four * A

    return far(t + one), near(t), near(one - t), far(two - t)


def time_resize_bicubic(x, out_len):
    """x [B,Tin,F] -> [B,out_len,F]: F.interpolate(x[:,None], (out_len, F), mode="bicubic", align_corners=True) restated. The
    feature axis keeps its size, so with align_corners its taps are exactly (0,1,0,0): a 1-D cubic along time. Source position
    of output row i is i*(Tin-1)/(out_len-1); taps are clamped at the borders (augment.py:138-149)."""
    import numpy as np
    xn = x.detach().cpu().numpy().astype(np.float32)
    Tin = xn.shape[1]
    scale = np.float32(Tin - 1) / np.float32(out_len - 1) if out_len > 1 else np.float32(0)
    src = scale * np.arange(out_len, dtype=np.float32)
    i0 = np.minimum(np.floor(src).astype(np.int64), Tin - 1)
    t = np.clip(src - i0.astype(np.float32), 0, 1)
    w = _cubic_taps(t)
    out = None
    for k in range(4):
        idx = np.clip(i0 - 1 + k, 0, Tin - 1)
        term = w[k][None, :, None] * xn[:, idx, :]
        out = term if out is None else out + term
    return torch.from_numpy(out.astype(np.float32))


def spec_augment(x, c=None, w=None, flen=None, fpos=None, tlen=None, tpos=None, window=5, replace_with_zero=False):
    """SpecAugment.forward with the random draws given: ``c`` = warp centre, ``w`` = its new position (both None = no warp),
    ``flen/fpos`` [B,n_freq_mask] and ``tlen/tpos`` [B,n_time_mask] = mask widths / starts (None = that mask is off).
    Order as the reference: warp, frequency masks (fill = mean of the warped tensor), time masks (fill = mean of the
    frequency-masked tensor) - augment.py:106-114,151-199."""
    x = x.clone().float()
    B, T, Fd = x.shape
    if c is not None and T - window > window:
        c, w = int(c), int(w)
        x = torch.cat([time_resize_bicubic(x[:, :c], w), time_resize_bicubic(x[:, c:], T - w)], dim=1)
    for lens, pos, dim in ((flen, fpos, 2), (tlen, tpos, 1)):
        if lens is None:
            continue
        D = x.shape[dim]
        lens, pos = torch.as_tensor(lens).view(B, -1, 1), torch.as_tensor(pos).view(B, -1, 1)
        ar = torch.arange(D).view(1, 1, -1)
        mask = ((pos <= ar) & (ar < pos + lens)).any(dim=1)
        mask = mask.unsqueeze(2) if dim == 1 else mask.unsqueeze(1)
        val = 0.0 if replace_with_zero else float(x.mean())
        x = x.masked_fill(mask, val)
    return x


# ----------------------------------------------------------------------------------------------
# A18  SpeedPerturb / Resample  (SB/processing/speech_augmentation.py:435-820; recipe: speeds [95,100,105] at 16 kHz,
#      conformer-t_scratch.yaml:144-146; applied to the mixture waveform, train_librispeechmix_scratch.py:82-85)
# ----------------------------------------------------------------------------------------------
def resample_filters(orig_freq, new_freq, lowpass_filter_width=6):
    """(first_indices [P] float, weights [P,W] float32): the polyphase windowed-sinc bank of Resample._indices_and_weights
    (:758-820). P = new/gcd output samples per unit of orig/gcd input samples; cutoff 0.99 * Nyquist of the lower rate,
    Hann window of half-width lowpass_filter_width / (2 cutoff)."""
    base = math.gcd(orig_freq, new_freq)
    P = new_freq // base
    cutoff = 0.99 * 0.5 * min(orig_freq, new_freq)
    half = lowpass_filter_width / (2.0 * cutoff)
    out_t = torch.arange(0.0, P) / new_freq
    lo = torch.ceil((out_t - half) * orig_freq)
    hi = torch.floor((out_t + half) * orig_freq)
    width = int((hi - lo + 1).max())
    idx = lo[:, None] + torch.arange(width)[None, :]
    dt = idx / orig_freq - out_t[:, None]
    wts = torch.zeros_like(dt)
    inside = dt.abs() < half
    wts[inside] = 0.5 * (1 + torch.cos(2 * math.pi * cutoff / lowpass_filter_width * dt[inside]))
    nz = dt != 0
    wts[nz] *= torch.sin(2 * math.pi * cutoff * dt[nz]) / (math.pi * dt[nz])
    wts[~nz] *= 2 * cutoff
    return lo, wts / orig_freq


def resample_out_len(n_in, orig_freq, new_freq):
    """Resample._output_samples (:705-756): number of output instants k/new_freq inside [0, n_in/orig_freq)."""
    if n_in <= 0:
        return 0
    tick = orig_freq * new_freq // math.gcd(orig_freq, new_freq)
    span, per_out = n_in * (tick // orig_freq), tick // new_freq
    last = span // per_out
    if last * per_out == span:
        last -= 1
    return last + 1


def resample(wav, orig_freq, new_freq, lowpass_filter_width=6):
    """wav [B,L] -> [B,resample_out_len(L)]. Output sample n = q*P + i (phase i of unit q) is the dot product of filter i with
    the input starting at first_indices[i] + q*stride, zeros outside the signal (Resample._perform_resample :618-703, which
    reaches the same sums through one strided conv1d per phase scattered by a transposed conv)."""
    import numpy as np
    if orig_freq == new_freq:
        return wav
    base = math.gcd(orig_freq, new_freq)
    stride, P = orig_freq // base, new_freq // base
    first, wts = resample_filters(orig_freq, new_freq, lowpass_filter_width)
    first, wts = first.numpy().astype(np.int64), wts.numpy().astype(np.float32)
    x = wav.detach().cpu().numpy().astype(np.float32)
    B, L = x.shape
    n_out = resample_out_len(L, orig_freq, new_freq)
    n = np.arange(n_out)
    start = first[n % P] + (n // P) * stride
    W = wts.shape[1]
    y = np.zeros((B, n_out), np.float32)
    for j in range(W):                       # same tap order as a conv1d accumulating left to right
        pos = start + j
        ok = (pos >= 0) & (pos < L)
        y += np.where(ok[None, :], x[:, np.clip(pos, 0, L - 1)], np.float32(0)) * wts[n % P, j][None, :]
    return torch.from_numpy(y)
