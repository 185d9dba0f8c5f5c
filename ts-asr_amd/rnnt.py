"""Host mirror of the reference's joiner + head + transducer loss on top of the fused HIP kernels.

Reference interface kept (names, argument meaning, error behaviour):
  * ``Transducer_joint(joint="sum", nonlinearity=LeakyReLU)``  speechbrain/nnet/transducer/transducer_joint.py:14-95
  * ``transducer_loss(logits, targets, input_lens, target_lens, blank_index, reduction, use_torchaudio)``
    speechbrain/nnet/losses.py:29-87
Extra (MI355X-native) entry: ``fused_joint_logits`` = joiner + transducer_head without the [B,T,U,J] tensor.
"""
import torch

from . import _capi as C
from . import prof

_LDL = 32  # logits rows are padded to 32 floats (128 B) so that every row access is a 16-byte multiple
JOINT_F32_EXACT = True   # fp32 activations: joint + head in exact fp32 arithmetic (csrc/joint_f32.hip). False: the MFMA kernels with fp32
                         # storage and bf16-rounded operands (tests that cover that instantiation switch it off)


def _exact(enc, J, V):
    return JOINT_F32_EXACT and enc.dtype == torch.float32 and J <= 768 and V <= 32


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class _JointLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, dec, weight, bias, slope, tlen, ulen):
        C.require_gpu(enc, dec, weight, bias)
        if enc.dtype != dec.dtype:
            raise ValueError("Arg 1 and 2 must have the same dtype")
        enc, dec = enc.contiguous(), dec.contiguous()
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        B, T, J = enc.shape
        U1 = dec.shape[1]
        V = w32.shape[0]
        buf = torch.empty(B, T, U1, _LDL, dtype=torch.float32, device=enc.device)
        ctx.exact = _exact(enc, J, V)
        with prof.region("joint_fwd"):
            if ctx.exact:
                C.check(C.lib().tsasr_joint_f32_fwd(C.ptr(enc), C.ptr(dec), C.ptr(w32), C.ptr(b32), C.ptr(buf), B, T, U1, J, V, _LDL,
                                                    float(slope), C.stream_ptr()), "tsasr_joint_f32_fwd")
            else:
                C.check(C.lib().tsasr_joint_fwd(C.ptr(enc), C.ptr(dec), C.ptr(w32), C.ptr(b32), C.ptr(buf), B, T, U1, J, V, _LDL,
                                                C.io_dtype(enc), float(slope), C.stream_ptr()), "tsasr_joint_fwd")
        ctx.save_for_backward(enc, dec, w32, tlen, ulen)
        ctx.slope, ctx.V = float(slope), V
        ctx.params = (weight, bias)
        return buf[..., :V]

    @staticmethod
    def backward(ctx, dlogits):
        enc, dec, w32, tlen, ulen = ctx.saved_tensors
        B, T, J = enc.shape
        U1, V = dec.shape[1], ctx.V
        dl = _as_padded_rows(dlogits, V)
        denc, ddec = torch.empty_like(enc), torch.empty_like(dec)
        dW = torch.empty(V, J, dtype=torch.float32, device=enc.device)
        db = torch.empty(V, dtype=torch.float32, device=enc.device)
        nws = C.lib().tsasr_joint_f32_bwd_workspace_bytes(B, U1, J) if ctx.exact else C.lib().tsasr_joint_bwd_workspace_bytes(B, T, U1, J)
        ws = _ws(nws, enc.device)
        with prof.region("joint_bwd"):
            if ctx.exact:
                C.check(C.lib().tsasr_joint_f32_bwd(C.ptr(dl), C.ptr(enc), C.ptr(dec), C.ptr(w32), C.ptr(denc), C.ptr(ddec), C.ptr(dW), C.ptr(db),
                                                    C.ptr(tlen), C.ptr(ulen), B, T, U1, J, V, dl.stride(-2), ctx.slope,
                                                    C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_joint_f32_bwd")
            else:
              C.check(C.lib().tsasr_joint_bwd(C.ptr(dl), C.ptr(enc), C.ptr(dec), C.ptr(w32), C.ptr(denc), C.ptr(ddec), C.ptr(dW), C.ptr(db),
                                            C.ptr(tlen), C.ptr(ulen), B, T, U1, J, V, dl.stride(-2), C.io_dtype(enc), ctx.slope,
                                            C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_joint_bwd")
        from .ops import _keep, _pgrad
        _keep(dW, db, ws)     # while reductions are deferred the two are filled from the slabs in `ws` by the batched launch at the end of backward
        return denc, ddec, _pgrad(ctx.params[0], dW), _pgrad(ctx.params[1], db), None, None, None


def _as_padded_rows(x, V):
    """A float32 [B,T,U1,V] tensor whose rows are 16-byte aligned multiples of 4 floats (copying only if needed)."""
    ldl = x.stride(-2) if x.dim() == 4 else 0
    ok = (x.dtype == torch.float32 and x.dim() == 4 and x.stride(-1) == 1 and ldl % 4 == 0 and ldl >= V
          and (ldl <= 32 or V > 32) and x.stride(1) == x.shape[2] * ldl and x.stride(0) == x.shape[1] * x.stride(1)
          and x.data_ptr() % 16 == 0)
    if ok:
        return x
    buf = torch.zeros(*x.shape[:3], _LDL if V <= _LDL else (V + 3) // 4 * 4, dtype=torch.float32, device=x.device)
    buf[..., :V] = x
    return buf[..., :V]


def fused_joint_logits(enc_proj, dec_proj, head_weight, head_bias, slope=0.01, enc_abs_lens=None, tok_abs_lens=None):
    """logits[b,t,u,:] = head(LeakyReLU(enc_proj[b,t] + dec_proj[b,u]))  as a [B,T,U1,V] view of 128-byte rows."""
    return _JointLogitsFn.apply(enc_proj, dec_proj, head_weight, head_bias, slope, enc_abs_lens, tok_abs_lens)


_LATTICE_ERR = []    # time-out words of the most recent split-lattice launches (csrc/rnnt.hip AB_SPIN_LIMIT)


def lattice_timeouts():
    """Number of kept split-lattice launches whose inter-workgroup wait ran out (a host read): their costs came out NaN."""
    return sum(int(w.view(torch.int32)[0].item() == 1) for w in _LATTICE_ERR)


class _RnntLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, tlen, ulen, blank):
        C.require_gpu(logits, targets, tlen, ulen)
        B, T, U1, V = logits.shape
        lg = _as_padded_rows(logits.float() if logits.dtype != torch.float32 else logits, V)
        tg = targets.to(torch.int32).contiguous()
        tlen, ulen = tlen.to(torch.int32).contiguous(), ulen.to(torch.int32).contiguous()
        if tg.dim() != 2 or tg.shape[0] != B or tg.shape[1] < U1 - 1:
            raise ValueError(f"targets must be [B, >= U1-1] = [{B}, >= {U1 - 1}], got {tuple(tg.shape)}")
        costs = torch.empty(B, dtype=torch.float32, device=lg.device)
        ws = _ws(C.lib().tsasr_rnnt_loss_workspace_bytes(B, T, U1), lg.device)
        with prof.region("rnnt_loss_fwd"):
            C.check(C.lib().tsasr_rnnt_loss_fwd(C.ptr(lg), C.ptr(tg), tg.stride(0), C.ptr(tlen), C.ptr(ulen), C.ptr(costs),
                                                B, T, U1, V, lg.stride(-2), int(blank), C.ptr(ws), ws.numel(), C.stream_ptr()),
                    "tsasr_rnnt_loss_fwd")
        off = int(C.lib().tsasr_rnnt_loss_error_word_offset(B, T, U1))
        if off >= 0:        # a lattice split over workgroups (long targets in small batches): keep its time-out word for Brain.flush_nonfinite
            _LATTICE_ERR.append(ws[off:off + 4])
            del _LATTICE_ERR[:-4]
        ctx.save_for_backward(lg, tg, tlen, ulen, ws)
        ctx.blank, ctx.in_dtype = int(blank), logits.dtype
        return costs

    @staticmethod
    def backward(ctx, gcosts):
        lg, tg, tlen, ulen, ws = ctx.saved_tensors
        B, T, U1, V = lg.shape
        ldl = lg.stride(-2)
        gs = gcosts.float().contiguous()
        buf = torch.empty(B, T, U1, ldl, dtype=torch.float32, device=lg.device)
        with prof.region("rnnt_loss_bwd"):
            C.check(C.lib().tsasr_rnnt_loss_bwd(C.ptr(lg), C.ptr(tg), tg.stride(0), C.ptr(tlen), C.ptr(ulen), C.ptr(gs), C.ptr(buf),
                                                B, T, U1, V, ldl, ctx.blank, C.ptr(ws), ws.numel(), C.stream_ptr()),
                    "tsasr_rnnt_loss_bwd")
        g = buf[..., :V]
        return (g if ctx.in_dtype == torch.float32 else g.to(ctx.in_dtype)), None, None, None, None


def rnnt_costs(logits, targets, abs_input_lens, abs_target_lens, blank=0):
    """Per-utterance -log P(y|x) with gradient w.r.t. logits (torchaudio.functional.rnnt_loss, reduction="none")."""
    return _RnntLossFn.apply(logits, targets, abs_input_lens, abs_target_lens, blank)


def transducer_loss(logits, targets, input_lens, target_lens, blank_index, reduction="mean", use_torchaudio=True):
    """Drop-in for speechbrain.nnet.losses.transducer_loss (losses.py:29-87).

    ``input_lens`` / ``target_lens`` are RELATIVE lengths; absolute = (rel * dim).round().int() exactly as
    losses.py:58-59. ``use_torchaudio`` selects the reference's two conventions: True (the recipes' default)
    = torchaudio semantics (no division by T); False = the Numba kernel's convention cost/T
    (speechbrain/nnet/loss/transducer_loss.py:104-106). Both run on the same HIP kernels.
    """
    from .nnet import abs_lengths_round
    tl = abs_lengths_round(input_lens, logits.shape[1])
    ul = abs_lengths_round(target_lens, targets.shape[1])
    costs = rnnt_costs(logits, targets, tl, ul, blank_index)
    if not use_torchaudio:
        costs = costs / tl.to(costs.dtype)
    if reduction == "mean":
        return costs.mean()
    if reduction == "sum":
        return costs.sum()
    if reduction == "none":
        return costs
    raise Exception("Unexpected reduction {}".format(reduction))


class Transducer_joint(torch.nn.Module):
    """Same constructor/forward as the reference joiner (transducer_joint.py:14-95) for joint="sum".

    ``forward(input_TN [B,T,1,J], input_PN [B,1,U,J])`` returns a lazy handle when the recipe immediately feeds
    the head (see ``FusedJointHead``); called stand-alone (e.g. by the searchers, one step at a time on
    [B,1,1,J]) it materialises nonlinearity(TN + PN) with ordinary device ops.
    """

    def __init__(self, joint_network=None, joint="sum", nonlinearity=torch.nn.LeakyReLU):
        super().__init__()
        if joint != "sum":
            raise NotImplementedError("ts-asr_amd implements joint='sum' (the only mode the TS-ASR recipes use)")
        self.joint_network = joint_network
        self.joint = joint
        self.nonlinearity = nonlinearity()

    def forward(self, input_TN, input_PN):
        if len(input_TN.shape) != len(input_PN.shape):
            raise ValueError("Arg 1 and 2 must be have same size")
        return self.nonlinearity(input_TN + input_PN)
