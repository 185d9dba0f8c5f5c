"""Operator layer between the host modules (nnet.py / conformer.py) and the device.

Everything that computes on the path is a hand-written gfx950 kernel behind the C-ABI (autograd.Functions in this file / rnnt.py);
``STATUS`` is the machine-readable list. Since round 4 no library kernel (MIOpen / hipBLASLt / rocFFT) is left on the path for the shapes
the three recipes use: the fp32 parity mode runs exact-fp32 HIP kernels (gemm_f32, attention_f32, joint_f32, lstm_f32), the searchers'
step-wise predictor calls run lstm_f32. A library module is only reached by LSTM shapes no kernel takes (several layers, bidirectional).
"""
import math
import os

import torch
import torch.nn.functional as F

from . import _capi as C
from . import prof

STATUS = {
    "joint_logits": "HIP", "rnnt_loss": "HIP", "layer_norm": "HIP", "bias_act_dropout": "HIP", "dropout_add": "HIP",
    "convmod_core": "HIP",
    "frontend_c1": "HIP", "frontend_im2col/col2im": "HIP (fp32 / other channel counts)",
    "frontend block 1 (bf16)": "HIP (frontend_block.hip: convolutions and filter gradients on MFMA)",
    "frontend block 2 (bf16, C_out = 128)": "HIP implicit GEMMs (conv_s2_fwd / conv_s2_wgrad / conv_s2_dgrad)",
    "matmul(bf16)": "HIP (gemm_bf16: fwd, dgrad, wgrad-into-arena)", "matmul(fp32 parity mode)": "HIP (gemm_f32: fp32 matrix cores; fwd, dgrad, wgrad)", "lstm(bf16 training)": "HIP persistent whole-sequence kernels (per-step kernels for other H)", "lstm(decoding / fp32 parity)": "HIP (lstm_f32: exact fp32, optional initial state)",
    "attention(fp32 parity, cross_attention injection)": "HIP (attention_f32: exact fp32)", "joint(fp32 parity)": "HIP (joint_f32)",
    "injection sum / prod": "HIP (inject)", "mix_sources": "HIP (dataio.hip)", "fbank": "HIP", "sentence_norm": "HIP",
    "relpos_attention": "HIP (forward: everything-in-LDS kernel for T <= 256 and, per chunk of 256 keys, for long sequences in small batches, streaming kernel otherwise; backward = query-major, key-major, d(pk) and partial-sum kernels)",
    # What is NOT hand-written: shapes / modes none of the three recipes reaches. Each of these routes goes through lib_fallback(): counted in
    # LIB_FALLBACKS, announced once (warnings), refused with TsasrHipMissing when ops.STRICT_HIP (env TSASR_STRICT_HIP=1) is set.
    # tests/test_recipe_gpu.py asserts that a fit + evaluate of every recipe YAML leaves LIB_FALLBACKS empty.
    "matmul(other dtypes / non-contiguous shapes no HIP GEMM takes)": "LIB (F.linear -> hipBLASLt), logged",
    "frontend conv block 2 (fp32 with the fp32 HIP GEMM switched off)": "LIB (F.linear), logged",
    "lstm(num_layers > 1, bidirectional, H not a multiple of 16 / > 1024)": "LIB (nn.LSTM -> MIOpen), logged",
    "cross_attention injection with head dim > 64": "LIB (ATen matmul / softmax), logged",
    "relpos_attention(need_weights=True: attention maps for plots)": "LIB (ATen matmul / softmax), logged",
}

STRICT_HIP = os.environ.get("TSASR_STRICT_HIP", "0") == "1"
LIB_FALLBACKS = {}      # route -> number of calls that left the hand-written path in this process


def lib_fallback(route, why=""):
    """A call is about to run a library kernel instead of a hand-written one: count it, say so once, refuse it in strict mode."""
    import warnings
    if STRICT_HIP:
        raise C.TsasrHipMissing(f"{route}: no hand-written gfx950 kernel takes this call ({why}); TSASR_STRICT_HIP=1 refuses the library route")
    if route not in LIB_FALLBACKS:
        warnings.warn(f"ts-asr_amd: {route} runs a library kernel, not a hand-written one ({why}); listed as LIB in ops.STATUS", RuntimeWarning, stacklevel=3)
    LIB_FALLBACKS[route] = LIB_FALLBACKS.get(route, 0) + 1

_seed_counter = [0]
_seed_dev = {}


def next_seed():
    """Per-call dropout stream id: torch's global seed mixed with the index of the call WITHIN the step. The per-step part lives
    in device memory (``seed_state``) and is advanced by ``begin_step`` - so the ids may be frozen into a captured hipGraph."""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B1 + _seed_counter[0] * 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF


def seed_state(device):
    """Device-resident uint64 step counter added to every dropout seed by the kernels (int64 storage, wraps harmlessly)."""
    key = str(device)
    if key not in _seed_dev:
        _seed_dev[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _seed_dev[key]


_LEN_CACHE = {}       # abs_lengths results of the current step


def begin_step(device):
    """Call once at the start of a training step: restarts the per-step call index and advances the device seed counter."""
    _seed_counter[0] = 0
    _LEN_CACHE.clear()
    if torch.device(device).type == "cuda":
        C.check(C.lib().tsasr_seed_advance(C.ptr(seed_state(device)), 0x9E3779B97F4A7C15, C.stream_ptr()), "tsasr_seed_advance")


# Job tables of the batched launches (csrc/reduce.hip, csrc/wgrad.hip, tsasr_accumulate_many): the host fills a PINNED table, an
# asynchronous copy moves it to a device table, the kernel reads that. Both copies are read when the stream gets there, not when the
# host enqueues them, so a pair may only be rewritten once the launch that used it has finished: eager launches rotate through a ring
# of pairs, each guarded by an event (the host waits only when it laps the GPU by a whole ring); a flush inside a stream capture gets
# a pair of its own for the life of the graph (filled at capture, uploaded once right after it - a replay carries no memcpy node).
class TableRing:
    def __init__(self, nbytes, device, eager_pairs=8, captured_pairs=8):
        self.nbytes, self.device = int(nbytes), torch.device(device)
        n = eager_pairs + captured_pairs
        self.host = [torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(n)]
        self.dev = [torch.empty(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(n)]
        self.eager_pairs, self.events, self.next_eager, self.captured, self.to_upload = eager_pairs, [None] * eager_pairs, 0, 0, []

    def acquire(self):
        """(index, host table, device table, captured?) for the flush being issued now."""
        if torch.cuda.is_current_stream_capturing():
            k = self.eager_pairs + self.captured
            if k >= len(self.host):
                raise RuntimeError("more captured flushes than job-table pairs (ops.TableRing captured_pairs)")
            self.captured += 1
            return k, self.host[k], self.dev[k], True
        k = self.next_eager
        self.next_eager = (k + 1) % self.eager_pairs
        if self.events[k] is not None:
            self.events[k].synchronize()   # the launch that last read this pair has finished (no-op unless the host is a ring ahead)
        return k, self.host[k], self.dev[k], False

    def reserve_captured(self, n):
        """Make sure the next capture can take ``n`` pairs: called OUTSIDE a capture (pinned and device memory cannot be allocated
        inside one), before every step capture - the pool grows with the number of captured graphs instead of being a fixed 120."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("TableRing.reserve_captured called inside a stream capture")
        while len(self.host) - (self.eager_pairs + self.captured) < n:
            self.host.append(torch.empty(self.nbytes, dtype=torch.uint8).pin_memory())
            self.dev.append(torch.empty(self.nbytes, dtype=torch.uint8, device=self.device))

    def launched(self, k, nbytes=None):
        """Call right after the launch that reads pair k was enqueued."""
        if k >= self.eager_pairs:
            self.to_upload.append((k, self.nbytes if nbytes is None else int(nbytes)))
            return
        ev = self.events[k] or torch.cuda.Event()
        ev.record()
        self.events[k] = ev

    def upload_captured(self):
        for k, nb in self.to_upload:
            self.dev[k][:nb].copy_(self.host[k][:nb], non_blocking=True)
        self.to_upload = []


# Deferred reductions (csrc/reduce.hip): between GradArena.begin_backward and finish_backward the partial-sum reductions of
# parameter gradients are queued and run as one launch; their workspaces must outlive the queue, so they are parked here.
_DEFER = {"on": False, "keep": [], "ring": None}
_DEFER_MAX_JOBS = 4096


def _ws(nbytes, device):
    t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    if _DEFER["on"]:
        _DEFER["keep"].append(t)
    return t


def _keep(*tensors):
    """Outputs of a queued reduction must not be recycled by the allocator before the flush writes them."""
    if _DEFER["on"]:
        _DEFER["keep"].extend(t for t in tensors if t is not None)


def reduce_defer_prepare(device):
    """Allocate the job tables (pinned host + device) outside of any graph capture."""
    if torch.device(device).type == "cuda":
        if _DEFER["ring"] is None:
            _DEFER["ring"] = TableRing(C.lib().tsasr_reduce_table_bytes(_DEFER_MAX_JOBS), device)
        if _WG["ring"] is None:
            _WG["ring"] = TableRing(C.lib().tsasr_wgrad_table_bytes(_WG_MAX_JOBS), device)


def reduce_defer_begin(device):
    if device.type != "cuda":
        return
    reduce_defer_prepare(device)
    C.check(C.lib().tsasr_reduce_defer(1), "tsasr_reduce_defer")
    _DEFER["on"] = True


def _reduce_flush(fn_name):
    n = C.lib().tsasr_reduce_pending()
    if n > _DEFER_MAX_JOBS:
        raise C.TsasrHipError("more queued reductions than the job table holds")
    if n > 0:
        ring = _DEFER["ring"]
        k, host, dev, _ = ring.acquire()
        C.check(getattr(C.lib(), fn_name)(C.ptr(host), C.ptr(dev), host.numel(), C.stream_ptr()), fn_name)
        ring.launched(k)


def reduce_flush():
    """Run the queued reductions now (one launch); the parked workspaces are released afterwards (stream order keeps them valid)."""
    if not _DEFER["on"]:
        return
    dpk_flush_joined()   # queued d(pk) passes read workspaces (dS, q + v, partials, key_lens) parked in _DEFER["keep"]: run them before the release
    _reduce_flush("tsasr_reduce_flush")
    _DEFER["keep"] = []


def upload_captured_tables():
    """After a stream capture: copy the job tables the captured flushes filled on the host to their device twins (a graph replay
    carries no memcpy node; each captured graph owns its pair for life)."""
    for st in (_DEFER, _WG, _DPK):
        if st["ring"] is not None:
            st["ring"].upload_captured()


def reserve_captured_tables(n):
    """Before a step capture: job-table pairs for up to ``n`` captured flushes of each batched launch (TableRing.reserve_captured)."""
    for st in (_DEFER, _WG, _DPK):
        if st["ring"] is not None:
            st["ring"].reserve_captured(n)


def discard_queues():
    """Error path of a failed step (a capture that raised half-way): drop every queued weight gradient and reduction, release their
    operands, leave deferral off - the next step starts from a clean state."""
    C.lib().tsasr_wgrad_discard()
    C.lib().tsasr_reduce_discard()
    C.lib().tsasr_relpos_dpk_discard()
    _DPK["on"] = False
    _DPK["outs"].clear()
    _WG["keep"], _WG["ids"], _WG["params"], _WG["flops"], _WG["tiles"], _WG["bytes"] = [], set(), [], 0.0, 0, 0.0
    _DEFER["keep"], _DEFER["on"] = [], False


def reduce_defer_end():
    if not _DEFER["on"]:
        return
    reduce_flush()
    C.check(C.lib().tsasr_reduce_defer(0), "tsasr_reduce_defer")
    _DEFER["on"] = False


# Grouped weight gradients (csrc/wgrad.hip): dW += dy^T . x feeds nothing downstream in backward, so _LinearFn / _FFNFn only QUEUE
# it while the gradient arena is collecting (operands parked here); GradArena flushes the queue in one launch per bucket / step.
_WG = {"keep": [], "ids": set(), "ring": None, "params": []}
_WG_MAX_JOBS = 1024
_WG_ENABLED = True
# The weight-gradient launch made beside another stream's kernels (the recipe's early flush) walks its tiles persistently on at most this
# many CUs: -1 = half the device. A 256x256-tile workgroup holds 128 KB of LDS, so nothing shares a CU with it; with one per tile the
# speaker branch's small kernels waited for whole CUs to drain (13.39-13.46 ms per step; 96 CUs 13.16, 112: 13.12, 128: 13.09, 144: 13.28,
# 192: 13.20). 0 = one workgroup per tile.
_WG_EARLY_WGS = int(os.environ.get("TSASR_WG_EARLY_WGS", "-1"))
_REDUCE_EARLY = os.environ.get("TSASR_REDUCE_EARLY", "1") != "0"
_WG_EARLY_SLOTS = 0      # (2: the launch made beside another stream's kernels uses 64 KB of LDS - measured equal)


def wgrad_queue(weight, grad2d, dy2, x2, key=None):
    """Queue grad2d [N, K] (fp32 view of weight.grad) += dy2[M, N]^T . x2[M, K]; False when the shapes do not fit the grouped kernel.
    ``key``: what identifies the OUTPUT tiles (default: the weight) - two gradients into disjoint column ranges of one weight (the `cat`
    injection's projection) pass distinct keys and share a launch; two into the same tiles are ordered by a flush."""
    key = id(weight) if key is None else key
    N, K = grad2d.shape
    M = dy2.shape[0]
    if not _WG_ENABLED or _WG["ring"] is None or N % 8 or K % 8 or N < 8 or K < 8 or dy2.stride(1) != 1 or x2.stride(1) != 1 \
            or dy2.stride(0) % 8 or x2.stride(0) % 8 or dy2.data_ptr() % 16 or x2.data_ptr() % 16 or grad2d.stride(1) != 1 \
            or dy2.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16 or grad2d.dtype != torch.float32:
        return False
    if key in _WG["ids"] or C.lib().tsasr_wgrad_pending() >= _WG_MAX_JOBS:   # a second gradient into the same tiles: order them
        # through the arena: the queue is shared by every stream of the step, so the launch has to be ordered after all of them and the
        # operands have to outlive the streams' join (GradArena.flush_wgrads, hold) - a bare launch on the current stream is neither
        sink = _GRAD_SINK
        if sink is not None and getattr(sink, "in_backward", False):
            sink.flush_wgrads(hold=True)
        else:
            wgrad_flush()
    C.check(C.lib().tsasr_wgrad_queue(C.ptr(dy2), C.ptr(x2), C.ptr(grad2d), N, K, M, dy2.stride(0), x2.stride(0), grad2d.stride(0)),
            "tsasr_wgrad_queue")
    _WG["ids"].add(key)
    _WG["keep"] += [dy2, x2]
    _WG["params"].append(weight)
    _WG["tiles"] = _WG.get("tiles", 0) + ((N + 255) // 256) * ((K + 255) // 256)
    return True


def wgrad_pending():
    return len(_WG["params"])


# Deferred d(pk) passes of the attention backward (csrc/attention.hip, tsasr_relpos_dpk_defer): queued per layer while the gradient arena
# is in backward, run as one grouped launch right in front of the grouped weight-gradient launch that consumes their outputs.
_DPK = {"ring": None, "on": False, "outs": set()}      # outs: data_ptr of every d(pk) tensor whose pass is still queued
_DPK_MAX_JOBS = 64
_DPK_DEFER = True        # (tests switch it off to compare with the per-layer launches)


def dpk_defer_begin(device):
    if not _DPK_DEFER or torch.device(device).type != "cuda":
        return
    if _DPK["ring"] is None:
        _DPK["ring"] = TableRing(C.lib().tsasr_relpos_dpk_table_bytes(_DPK_MAX_JOBS), device)
    C.check(C.lib().tsasr_relpos_dpk_defer(1), "tsasr_relpos_dpk_defer")
    _DPK["on"] = True


def dpk_deferring():
    return _DPK["on"]


def dpk_flush():
    """Run every queued d(pk) pass now, on the current stream (two launches); a no-op when nothing is queued."""
    _DPK["outs"].clear()
    if _DPK["ring"] is None or C.lib().tsasr_relpos_dpk_pending() == 0:
        return
    ring = _DPK["ring"]
    k, host, dev, _ = ring.acquire()
    with prof.region("relpos_dpk_group"):
        C.check(C.lib().tsasr_relpos_dpk_flush(C.ptr(host), C.ptr(dev), host.numel(), C.stream_ptr()), "tsasr_relpos_dpk_flush")
    ring.launched(k)


def dpk_flush_joined():
    """dpk_flush from inside a backward node (a reader needs a queued pass's output now): the C-side queue is shared by every stream of
    the step, so the launch is first ordered behind all of them (GradArena.join_streams) - a bare flush on the current stream could run
    a pass queued by the other branch's backward before its producers have finished."""
    if _DPK["ring"] is None or C.lib().tsasr_relpos_dpk_pending() == 0:
        _DPK["outs"].clear()
        return
    sink = _GRAD_SINK
    if sink is not None and getattr(sink, "in_backward", False) and hasattr(sink, "join_streams"):
        sink.join_streams()
    dpk_flush()


def dpk_defer_end():
    if not _DPK["on"]:
        return
    dpk_flush()
    C.check(C.lib().tsasr_relpos_dpk_defer(0), "tsasr_relpos_dpk_defer")
    _DPK["on"] = False


def wgrad_flush(hold=None):
    """One launch for every queued weight gradient (on the current stream, which must be ordered after their producers); returns the
    parameters whose gradients it completed. `hold`: a list that takes over the operand references (a launch on a side stream: the
    caller releases them once the consumer stream has joined it)."""
    done = _WG["params"]
    dpk_flush()      # queued d(pk) passes produce operands of the linear_pos weight gradients in this launch: they go first, same stream
    if hold is not None and _REDUCE_EARLY and _DEFER["on"]:
        # the early flush (beside another stream's backward): the partial-sum reductions queued so far go with it, IN FRONT of the long grouped
        # launch - the reduction left for the end of backward (on the step's critical path) then covers the rest of the step only. The parked
        # workspaces travel with the held operands (some were allocated on the other stream).
        _reduce_flush("tsasr_reduce_flush")
        hold.extend(_DEFER["keep"])
        _DEFER["keep"] = []
    if done:
        ring = _WG["ring"]
        k, host, dev, _ = ring.acquire()
        if hold is not None and _WG_EARLY_SLOTS:
            C.lib().tsasr_wgrad_next_flush_slots(_WG_EARLY_SLOTS)
        if hold is not None and _WG_EARLY_WGS:
            C.lib().tsasr_wgrad_next_flush_wgs(_WG_EARLY_WGS if _WG_EARLY_WGS > 0 else max(1, torch.cuda.get_device_properties(dev.device).multi_processor_count // 2))
        with prof.region("wgrad_group", _WG.get("flops", 0.0), _WG.get("bytes", 0.0)):
            C.check(C.lib().tsasr_wgrad_flush(C.ptr(host), C.ptr(dev), host.numel(), C.stream_ptr()), "tsasr_wgrad_flush")
        ring.launched(k)
        if hold is not None:
            hold.extend(_WG["keep"])
    _WG["keep"], _WG["ids"], _WG["params"], _WG["flops"], _WG["tiles"], _WG["bytes"] = [], set(), [], 0.0, 0, 0.0
    return done


def _f32(p):
    return p if p.dtype == torch.float32 else p.float()


def _w(p, like):
    """Parameter in the activation dtype (fp32 master weights stay untouched; autograd casts the gradient back)."""
    return p if p is None or p.dtype == like.dtype else p.to(like.dtype)


# ---------------------------------------------------------------------------------------------------------
_GRAD_SINK = None


def set_grad_sink(arena):
    """The gradient arena that weight-gradient GEMMs may accumulate into directly (dp.GradArena protocol)."""
    global _GRAD_SINK
    _GRAD_SINK = arena


def _pgrad(param, g, shape=None):
    """Return value for a parameter gradient slot of an autograd.Function.backward: None when the gradient arena takes the
    fp32 temporary itself (one batched add at the end of backward), else the tensor autograd should accumulate."""
    if g is None or param is None:
        return None
    sink = _GRAD_SINK
    if sink is not None and isinstance(param, torch.nn.Parameter) and sink.defer_add(param, g):
        return None
    reduce_flush()   # autograd will read g right away: a queued reduction that produces it has to run first
    g = g.view(shape if shape is not None else param.shape)
    return g if g.dtype == param.dtype else g.to(param.dtype)


def gemm_bf16(a, b, M, N, K, lda, ldb, trans_a, trans_b, out=None, out_dtype=torch.bfloat16, accumulate=False, ldc=None, defer_ok=False):
    """C[M,N] (+)= op(A).op(B) on the hand-written MFMA kernel (csrc/gemm.hip); see include/tsasr_hip.h for the layouts."""
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    elif ldc is None:     # a caller-owned output may be a column range of a wider matrix: its row stride, not N
        if out.dim() != 2 or out.stride(1) != 1:
            raise ValueError("gemm_bf16: `out` must be a 2-D tensor with unit column stride")
        ldc = out.stride(0)
    od = C.F32 if out.dtype == torch.float32 else C.BF16
    nws = C.lib().tsasr_gemm_bf16_workspace_bytes(M, N, K, od)
    ws = _ws(nws, a.device) if nws else None
    if prof.ENABLED:  # label = the kernel template instantiation rocprofv3 will show (mirror of plan() in csrc/gemm.hip)
        t0, t1, t2 = ((M + 127) // 128) * ((N + 127) // 128), ((M + 127) // 128) * ((N + 63) // 64), ((M + 63) // 64) * ((N + 63) // 64)
        tiles, tile = (t0, "128, 128") if t0 >= 512 else ((t1, "128, 64") if t1 >= 192 else (t2, "64, 64"))
        nsplit = min(32, max(1, K // 128), (768 + tiles - 1) // tiles) if (od == C.F32 and tiles < 256 and K >= 384) else 1
        kchunk = -(-(-(-K // nsplit)) // 64) * 64
        mode = 1 if (nsplit > 1 or (od == C.F32 and not accumulate)) else (2 if accumulate else 0)
        ring = K % 64 == 0 and (tile != "128, 128" or min(K, kchunk) >= 1024) and (M >= 8 or not trans_a) and (N >= 8 or not trans_b)
        if ring and tile == "64, 64" and trans_a and trans_b and mode != 0:
            label = f"gemm_tt64_wavek_kernel<{mode}>"
        elif ring and tile == "64, 64" and not trans_a and not trans_b and mode == 0 and min(K, kchunk) >= 1024:
            label = "gemm_nn64_wavek_kernel"
        else:
            label = (f"gemm_bf16_{'ring_' if ring else ''}kernel<{tile}, {'true' if trans_a else 'false'}, "
                     f"{'true' if trans_b else 'false'}, {mode}{', 3' if ring else ''}>")       # (ring kernels: + the slot count)
    else:
        label = "gemm"
    with prof.region(label, 2.0 * M * N * K):
        C.check(C.lib().tsasr_gemm_bf16(C.ptr(a), C.ptr(b), C.ptr(out), M, N, K, lda, ldb, N if ldc is None else ldc, int(trans_a),
                                        int(trans_b), od, (2 if defer_ok and _DEFER["on"] else 1) if accumulate else 0,
                                        C.ptr(ws), 0 if ws is None else ws.numel(), C.stream_ptr()),
                "tsasr_gemm_bf16")
    return out


def _bf16_weight(w):
    sh = getattr(w, "_bf16", None)
    if sh is not None and getattr(w, "_bf16_ver", -1) == w._version:
        return sh
    return w.detach().to(torch.bfloat16)


def _bf16_weight_t(w):
    """Transposed bf16 copy [in, out] of a Linear weight kept by the gradient arena (dp.GradArena.refresh_transposed), or None."""
    sh = getattr(w, "_bf16_t", None)
    if sh is not None and getattr(w, "_bf16_ver", -1) == w._version and sh.shape[0] % 8 == 0 and sh.shape[1] % 8 == 0:
        return sh
    return None


def _dgrad(dy2, weight, w16, M, N, K):
    """dx [M, K] = dy [M, N] . W [N, K]: through the transposed copy W^T [K, N] when the arena keeps one (both operands
    k-contiguous - the forward GEMM's fast path), else through the transposing fragment reads."""
    wt = _bf16_weight_t(weight)
    if wt is not None:
        return gemm_bf16(dy2, wt, M, K, N, N, N, 0, 0)
    return gemm_bf16(dy2, w16, M, K, N, N, w16.stride(0), 0, 1)


def _gemm_ok(x, weight):
    if weight.dim() == 3 and weight.shape[2] == 1 and weight.is_contiguous():   # kernel-size-1 Conv1d weight [N, K, 1]: the same matrix
        return (x.dtype == torch.bfloat16 and x.is_cuda and weight.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0
                and x.shape[-1] == weight.shape[1])
    return (x.dtype == torch.bfloat16 and x.is_cuda and weight.dim() == 2 and weight.stride(1) == 1 and weight.shape[1] % 8 == 0
            and weight.shape[0] % 8 == 0 and weight.stride(0) % 8 == 0 and x.shape[-1] == weight.shape[1])


class _LinearFn(torch.autograd.Function):
    """y = x . W^T on the HIP GEMM; dgrad on the same kernel; the weight gradient is added straight into the fp32 gradient
    arena when one is registered (no bf16->fp32 cast kernel, no separate accumulate kernel)."""

    @staticmethod
    def forward(ctx, x, weight):
        N, K = weight.shape[0], weight.shape[1]          # [N, K] or a kernel-size-1 Conv1d weight [N, K, 1] (the Parameter itself:
        x2 = x.reshape(-1, K)                            # its bf16 shadows and its slot in the gradient arena are found through it)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w16 = _bf16_weight(weight).view(N, K) if weight.dim() == 3 else _bf16_weight(weight)
        if w16.stride(1) != 1 or w16.stride(0) % 8 != 0 or w16.data_ptr() % 16 != 0:
            w16 = w16.contiguous()
        M = x2.shape[0]
        y = gemm_bf16(x2, w16, M, N, K, K, w16.stride(0), 0, 0)
        ctx.save_for_backward(x2, w16)
        ctx.weight = weight
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w16 = ctx.saved_tensors
        weight = ctx.weight
        N, K = weight.shape[0], weight.shape[1]
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(dy2, weight, w16, M, N, K).view(ctx.xshape)                         # dy . W
        dw = None
        if ctx.needs_input_grad[1]:
            sink = _GRAD_SINK
            if (sink is not None and weight.is_leaf and sink.accepts(weight) and weight.grad.dtype == torch.float32
                    and weight.grad.is_contiguous()):
                _wgrad_into(sink, weight, weight.grad.view(N, K), dy2, x2)                           # grad += dy^T . x
            else:
                if dy.data_ptr() in _DPK["outs"] or dy2.data_ptr() in _DPK["outs"]:
                    dpk_flush_joined()     # dy is a deferred d(pk): the plain GEMM reads it now
                dw = gemm_bf16(dy2, x2, N, K, M, N, K, 1, 1, out_dtype=torch.float32).to(weight.dtype).view(weight.shape)
        return dx, dw


class _LinearColsFn(torch.autograd.Function):
    """y = x . W[:, c0:c0+n]^T (+ bias): a Linear over a COLUMN RANGE of a weight - the two halves of the `cat` injection's projection
    (models/conformer.py:254-262: Linear(2D -> D) applied to [src | spk]; here src and spk are multiplied separately). Works on the
    Parameter itself: the operand is a strided view of its bf16 shadow (no cast of a slice), the weight gradient goes into the same
    columns of its slot in the gradient arena, queued with the other weight gradients (no slice-backward, no separate GEMM)."""

    @staticmethod
    def forward(ctx, x, weight, bias, c0, n):
        N, Kf = weight.shape
        x2 = x.reshape(-1, n)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w16 = _bf16_weight(weight)[:, c0:c0 + n]
        M = x2.shape[0]
        if bias is not None:
            y = gemm_bf16_fused(x2, w16, M, N, n, n, w16.stride(0), 0, 0, 1, bias=_f32(bias).contiguous())
        else:
            y = gemm_bf16(x2, w16, M, N, n, n, w16.stride(0), 0, 0)
        ctx.save_for_backward(x2, w16)
        ctx.weight, ctx.bias, ctx.cols, ctx.xshape = weight, bias, (int(c0), int(n)), x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w16 = ctx.saved_tensors
        weight, bias, (c0, n) = ctx.weight, ctx.bias, ctx.cols
        N, Kf = weight.shape
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        db = _pgrad(bias, colsum(dy2)) if (bias is not None and ctx.needs_input_grad[2]) else None
        dx = gemm_bf16(dy2, w16, M, n, N, N, w16.stride(0), 0, 1).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            sink = _GRAD_SINK
            if (sink is not None and weight.is_leaf and sink.accepts(weight) and weight.grad.dtype == torch.float32
                    and weight.grad.is_contiguous()):
                _wgrad_into(sink, weight, weight.grad.view(N, Kf)[:, c0:c0 + n], dy2, x2, key=(id(weight), c0))
            else:
                dw = torch.zeros(N, Kf, dtype=torch.float32, device=dy2.device)
                gemm_bf16(dy2, x2, N, n, M, N, n, 1, 1, out=dw[:, c0:c0 + n], ldc=Kf)
                dw = dw.to(weight.dtype)
        return dx, dw, db, None, None


def linear_cols_ok(x, weight, c0, n):
    return (x.dtype == torch.bfloat16 and x.is_cuda and weight.dim() == 2 and weight.is_contiguous() and x.shape[-1] == n
            and n % 8 == 0 and c0 % 8 == 0 and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0 and c0 + n <= weight.shape[1])


def linear_cols(x, weight, bias, c0, n):
    """x . W[:, c0:c0+n]^T (+ bias) on the HIP GEMM (see _LinearColsFn)."""
    return _LinearColsFn.apply(x, weight, bias, int(c0), int(n))


# device arrays of weight-shadow addresses for the batched projections, by the tuple of addresses. Kept for the life of the process: a
# captured step has the table's address baked into its batched-GEMM node (a few hundred bytes per distinct set of weights)
_PTR_TABLES = {}


class _ProjectManyFn(torch.autograd.Function):
    """(x . W_0^T, ..., x . W_{L-1}^T) for one shared x [M, K] and L same-shaped weights in ONE launch (tsasr_gemm_bf16_nt_batched): the
    projections of the positional table for every layer of an encoder. x carries no gradient; each weight gradient goes where
    _LinearFn.backward would send it (the arena's grouped launch)."""

    @staticmethod
    def forward(ctx, x2, table, *weights):
        L = len(weights)
        N, K = weights[0].shape
        M = x2.shape[0]
        out = torch.empty(L, M, N, dtype=torch.bfloat16, device=x2.device)
        with prof.region("gemm_bf16_nt_batched", 2.0 * M * N * K * L):
            C.check(C.lib().tsasr_gemm_bf16_nt_batched(C.ptr(x2), C.ptr(table), C.ptr(out), M, N, K, x2.stride(0), K, N, M * N, L,
                                                       C.stream_ptr()), "tsasr_gemm_bf16_nt_batched")
        ctx.save_for_backward(x2)
        ctx.weights = weights
        return tuple(out[i] for i in range(L))

    @staticmethod
    def backward(ctx, *dys):
        (x2,) = ctx.saved_tensors
        M, K = x2.shape
        grads = []
        for i, (w, dy) in enumerate(zip(ctx.weights, dys)):
            dw = None
            if dy is not None and ctx.needs_input_grad[2 + i]:
                N = w.shape[0]
                dy2 = dy.reshape(M, N)
                if not dy2.is_contiguous():
                    dy2 = dy2.contiguous()
                sink = _GRAD_SINK
                if sink is not None and w.is_leaf and sink.accepts(w) and w.grad.dtype == torch.float32 and w.grad.is_contiguous():
                    _wgrad_into(sink, w, w.grad.view(N, K), dy2, x2)
                else:
                    if dy.data_ptr() in _DPK["outs"] or dy2.data_ptr() in _DPK["outs"]:
                        dpk_flush_joined()
                    dw = gemm_bf16(dy2, x2, N, K, M, N, K, 1, 1, out_dtype=torch.float32).to(w.dtype).view(w.shape)
            grads.append(dw)
        return (None, None, *grads)


def project_many(x2, weights):
    """[x2 . W^T for W in weights] (bf16 [M, N] each) in one launch, or None when the batched kernel does not take the case: every
    weight needs a live bf16 shadow at a fixed address (the gradient arena's: the address table is built once per set of weights,
    outside any stream capture), same shape, K a multiple of 64."""
    if len(weights) < 2 or x2.dim() != 2 or not x2.is_cuda or x2.dtype != torch.bfloat16 or x2.requires_grad or x2.stride(1) != 1 \
            or x2.stride(0) % 8 or x2.data_ptr() % 16:
        return None
    N, K = weights[0].shape[0], weights[0].shape[1]
    if K % 64 or N % 8 or x2.shape[1] != K:
        return None
    ptrs = []
    for w in weights:
        sh = getattr(w, "_bf16", None)
        if (w.dim() != 2 or tuple(w.shape) != (N, K) or sh is None or getattr(w, "_bf16_ver", -1) != w._version or not sh.is_contiguous()
                or sh.data_ptr() % 16 or not w.is_leaf):
            return None
        ptrs.append(sh.data_ptr())
    key = tuple(ptrs)
    table = _PTR_TABLES.get(key)
    if table is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        table = _PTR_TABLES[key] = torch.tensor(ptrs, dtype=torch.int64).to(x2.device)
    return list(_ProjectManyFn.apply(x2, table, *weights))


class _LinearEpiFn(torch.autograd.Function):
    """dropout_p(LeakyReLU_slope(x . W^T + bias)) with the whole chain in the GEMM's epilogue (tsasr_gemm_bf16_fused mode 1: slope < 0 = no
    activation): no element-wise pass over the output. Backward: the mask / activation derivative and the bias gradient in one pass
    over dy when there is a mask or an activation (the kernel regenerates the mask from the seed), the bias gradient alone as column
    sums otherwise; then the data and weight gradients exactly as _LinearFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope, p, seed):
        N, K = weight.shape[0], weight.shape[1]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w16 = _bf16_weight(weight).view(N, K) if weight.dim() == 3 else _bf16_weight(weight)
        if w16.stride(1) != 1 or w16.stride(0) % 8 != 0 or w16.data_ptr() % 16 != 0:
            w16 = w16.contiguous()
        M = x2.shape[0]
        b = None if bias is None else _f32(bias).contiguous()
        y = gemm_bf16_fused(x2, w16, M, N, K, K, w16.stride(0), 0, 0, 1, bias=b, slope=float(slope), p=float(p), seed=seed)
        masked = slope >= 0 or p > 0
        ctx.save_for_backward(x2, w16, y if masked else None)
        ctx.weight, ctx.bias, ctx.xshape, ctx.cfg = weight, bias, x.shape, (float(slope), float(p), seed)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w16, y = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        slope, p, seed = ctx.cfg
        N, K = weight.shape[0], weight.shape[1]
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        db = None
        if y is not None:
            dz = torch.empty_like(y)
            db = torch.empty(N, dtype=torch.float32, device=y.device) if bias is not None else None
            _keep(db)
            ws = _ws(C.lib().tsasr_colpart_workspace_bytes(M, N), y.device) if bias is not None else None
            with prof.region("bias_act_dropout_bwd"):
                C.check(C.lib().tsasr_bias_act_dropout_bwd(C.ptr(dy2), C.ptr(y), C.ptr(dz), C.ptr(db), M, N, slope, p, seed,
                                                           C.ptr(seed_state(y.device)), C.io_dtype(y),
                                                           C.ptr(ws), 0 if ws is None else ws.numel(), C.stream_ptr()),
                        "tsasr_bias_act_dropout_bwd")
            dy2 = dz
        elif bias is not None and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(dy2, weight, w16, M, N, K).view(ctx.xshape)
        dw = None
        if ctx.needs_input_grad[1]:
            sink = _GRAD_SINK
            if (sink is not None and weight.is_leaf and sink.accepts(weight) and weight.grad.dtype == torch.float32
                    and weight.grad.is_contiguous()):
                _wgrad_into(sink, weight, weight.grad.view(N, K), dy2, x2)
            else:
                dw = gemm_bf16(dy2, x2, N, K, M, N, K, 1, 1, out_dtype=torch.float32).to(weight.dtype).view(weight.shape)
        return dx, dw, (None if db is None else _pgrad(bias, db)), None, None, None


def _wgrad_into(sink, weight, grad2d, dy2, x2, key=None):
    """grad2d [N, K] += dy2 [M, N]^T . x2 [M, K]: queued for the arena's grouped launch (csrc/wgrad.hip) when it collects them,
    else one split-K GEMM now."""
    N, K = grad2d.shape
    M = dy2.shape[0]
    if getattr(sink, "collect_wgrads", False) and wgrad_queue(weight, grad2d, dy2, x2, key):
        _WG["flops"] = _WG.get("flops", 0.0) + 2.0 * M * N * K
        _WG["bytes"] = _WG.get("bytes", 0.0) + 2.0 * M * (N + K) + 8.0 * N * K   # dy and x (bf16) read once, the fp32 gradient read and written
        sink.wgrad_queued(weight)
        return
    if dy2.data_ptr() in _DPK["outs"]:
        dpk_flush_joined()     # dy2 is a deferred d(pk): the split-K GEMM below reads it now
    # grad2d may be a column range of the weight's slot (_LinearColsFn): its rows are grad2d.stride(0) apart, and the slab-deferred
    # accumulate form (contiguous output only) is not for it
    gemm_bf16(dy2, x2, N, K, M, dy2.stride(0), x2.stride(0), 1, 1, out=grad2d, accumulate=True, ldc=grad2d.stride(0),
              defer_ok=grad2d.stride(0) == K)
    sink.mark_ready(weight)


def gemm_bf16_fused(a, b, M, N, K, lda, ldb, trans_a, trans_b, mode, bias=None, y=None, slope=-1.0, p=0.0, seed=0, dbias=None, mask=None):
    """``mask`` (uint16 [M, N/8], only where fused_mask_ok): written by the mode-1 call, read by the mode-2 call instead of y / the hash."""
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    ws = _ws(C.lib().tsasr_gemm_bf16_fused_workspace_bytes(M, N), a.device) if dbias is not None else None
    with prof.region(f"gemm_bf16_fused<{mode}>", 2.0 * M * N * K):
        C.check(C.lib().tsasr_gemm_bf16_fused(C.ptr(a), C.ptr(b), C.ptr(out), M, N, K, lda, ldb, N, int(trans_a), int(trans_b), int(mode),
                                              C.ptr(bias), C.ptr(y), 0 if y is None else y.stride(0), float(slope), float(p), seed,
                                              C.ptr(seed_state(a.device)), C.ptr(dbias), C.ptr(mask), C.ptr(ws), 0 if ws is None else ws.numel(),
                                              C.stream_ptr()), "tsasr_gemm_bf16_fused")
    return out


def fused_mask_ok(M, N, K):
    return bool(C.lib().tsasr_gemm_bf16_fused_mask_ok(int(M), int(N), int(K)))


_FFN_MASK = True      # (tests switch it off: the data gradient then re-reads the activation and re-hashes the keep-bits - same bits)


class _FFNFn(torch.autograd.Function):
    """PositionalwiseFeedForward core (SB/nnet/attention.py:820-836): Linear(D->F) + bias + LeakyReLU + Dropout + Linear(F->D),
    as two HIP GEMMs: the first with the bias/activation/dropout epilogue, the second plain. Backward: dgrad of the second
    GEMM carries the activation/dropout backward (and the bias-gradient column sums) in ITS epilogue; both weight gradients
    are added straight into the gradient arena. The [M, F] hidden activation is written once and read twice - no
    elementwise pass over it in either direction."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, slope, p, seed):
        F1, D = w1.shape
        x2 = x.reshape(-1, D)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        w1h, w2h = _bf16_weight(w1).contiguous(), _bf16_weight(w2).contiguous()
        b1f = None if b1 is None else _f32(b1).contiguous()
        # the epilogue's mask words (keep-bits + sign of the stored activation, 2 bits per element) are kept for the backward when both of
        # its GEMMs take them: the data gradient then reads M*F1/4 bytes instead of the whole activation (2*M*F1) and hashes nothing
        mask = None
        if _FFN_MASK and (slope >= 0 or p > 0) and slope <= 1 and F1 % 8 == 0 and fused_mask_ok(M, F1, D) and fused_mask_ok(M, F1, w2.shape[0]) \
                and _bf16_weight_t(w2) is not None:
            mask = torch.empty(M, F1 // 8, dtype=torch.int16, device=x2.device)
        h = gemm_bf16_fused(x2, w1h, M, F1, D, D, D, 0, 0, 1, bias=b1f, slope=slope, p=p, seed=seed, mask=mask)
        o = gemm_bf16(h, w2h, M, w2.shape[0], F1, F1, F1, 0, 0)
        ctx.save_for_backward(x2, h, w1h, w2h, mask)
        ctx.cfg = (float(slope), float(p), seed, (w1, b1, w2), x.shape)
        return o.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, do):
        x2, h, w1h, w2h, mask = ctx.saved_tensors
        slope, p, seed, (w1, b1, w2), xshape = ctx.cfg
        F1, D = w1.shape
        Dout = w2.shape[0]
        M = x2.shape[0]
        do2 = do.reshape(M, Dout)
        if not do2.is_contiguous():
            do2 = do2.contiguous()
        db1 = torch.empty(F1, dtype=torch.float32, device=x2.device) if b1 is not None else None
        _keep(db1)
        # dh_pre = (do . W2) * dropout/activation backward, + column sums -> db1
        w2t = _bf16_weight_t(w2)
        if w2t is not None:   # W2^T [F1, Dout]: k-contiguous operand
            dh = gemm_bf16_fused(do2, w2t, M, F1, Dout, Dout, Dout, 0, 0, 2, y=h, slope=slope, p=p, seed=seed, dbias=db1, mask=mask)
        else:
            dh = gemm_bf16_fused(do2, w2h, M, F1, Dout, Dout, F1, 0, 1, 2, y=h, slope=slope, p=p, seed=seed, dbias=db1)
        dx = _dgrad(dh, w1, w1h, M, F1, D).view(xshape) if ctx.needs_input_grad[0] else None
        sink = _GRAD_SINK

        def wgrad(w, g, a, n_out, k_in):
            if sink is not None and w.is_leaf and sink.accepts(w) and w.grad.dtype == torch.float32 and w.grad.is_contiguous():
                _wgrad_into(sink, w, w.grad, g, a)
                return None
            return gemm_bf16(g, a, n_out, k_in, M, n_out, k_in, 1, 1, out_dtype=torch.float32).to(w.dtype)

        dw2 = wgrad(w2, do2, h, Dout, F1)
        dw1 = wgrad(w1, dh, x2, F1, D)
        return dx, dw1, _pgrad(b1, db1), dw2, None, None, None


def ffn_core(x, w1, b1, w2, slope, p, training):
    """Linear(w1,b1) -> LeakyReLU -> Dropout -> Linear(w2) (bias of the second Linear is applied by the caller's fused tail)."""
    p = float(p) if training else 0.0
    if _gemm_ok(x, w1) and _gemm_ok(x.new_empty(0, w2.shape[1]), w2) and w1.shape[0] % 8 == 0:
        return _FFNFn.apply(x, w1, b1, w2, -1.0 if slope is None else slope, p, next_seed() if p > 0 else 0)
    return matmul_nt(linear(x, w1, b1, slope, p, training), w2)


def gemm_f32(a, b, M, N, K, lda, ldb, trans_a, trans_b, out=None, accumulate=False):
    """C[M,N] (+)= op(A).op(B) in fp32 on the hand-written fp32-MFMA kernel (csrc/gemm_f32.hip); layouts as gemm_bf16."""
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    with prof.region("gemm_f32", 2.0 * M * N * K):
        C.check(C.lib().tsasr_gemm_f32(C.ptr(a), C.ptr(b), C.ptr(out), M, N, K, lda, ldb, out.stride(0), int(trans_a), int(trans_b),
                                       int(bool(accumulate)), C.stream_ptr()), "tsasr_gemm_f32")
    return out


class _LinearF32Fn(torch.autograd.Function):
    """y = x . W^T in fp32 (the parity mode): forward, data gradient and weight gradient on tsasr_gemm_f32 - no library GEMM."""

    @staticmethod
    def forward(ctx, x, weight):
        N, K = weight.shape[0], weight.shape[1]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w2 = weight.detach().reshape(N, K)
        if not w2.is_contiguous():
            w2 = w2.contiguous()
        M = x2.shape[0]
        y = gemm_f32(x2, w2, M, N, K, K, K, 0, 0)
        ctx.save_for_backward(x2, w2)
        ctx.wshape, ctx.xshape = weight.shape, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w2 = ctx.saved_tensors
        N, K = w2.shape
        M = x2.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = gemm_f32(dy2, w2, M, K, N, N, K, 0, 1).view(ctx.xshape) if ctx.needs_input_grad[0] else None      # dy . W
        dw = gemm_f32(dy2, x2, N, K, M, N, K, 1, 1).view(ctx.wshape) if ctx.needs_input_grad[1] else None      # dy^T . x
        return dx, dw


_F32_HIP_GEMM = True      # fp32 (parity-mode) Linear layers on the hand-written fp32 GEMM; False: the library GEMM (tests compare the two)


def _gemm_f32_shapes_ok(x, weight):
    return (_F32_HIP_GEMM and x.is_cuda and weight.dtype == torch.float32 and weight.is_cuda
            and (weight.dim() == 2 or (weight.dim() == 3 and weight.shape[2] == 1)) and x.shape[-1] == weight.shape[1] and x.numel() > 0)


def _gemm_f32_ok(x, weight):
    return x.dtype == torch.float32 and _gemm_f32_shapes_ok(x, weight)


def _gemm_f32_widen_ok(x, weight):
    """bf16 activations against a matrix the bf16 GEMM does not take (N % 8 != 0: the 29-row transducer head the searchers call one lattice
    cell at a time, SB/decoders/transducer.py:375-384): small enough to widen to fp32 and run the fp32 HIP GEMM."""
    return x.dtype == torch.bfloat16 and x.numel() <= (1 << 16) and _gemm_f32_shapes_ok(x, weight)


def matmul_nt(x, weight):
    """x @ weight^T. bf16 activations: hand-written MFMA GEMM (csrc/gemm.hip); fp32 activations (parity runs): the hand-written fp32
    GEMM (csrc/gemm_f32.hip: fp32 matrix cores, operands and sums in fp32)."""
    if _gemm_ok(x, weight):
        return _LinearFn.apply(x, weight)
    if _gemm_f32_ok(x, weight):
        return _LinearF32Fn.apply(x, weight)
    if _gemm_f32_widen_ok(x, weight):     # (the weight rounded to bf16 first: the operand every other bf16 GEMM of the mode sees - its bf16 shadow)
        return _LinearF32Fn.apply(x.float(), weight.to(torch.bfloat16).float()).to(x.dtype)
    if weight.dim() == 3:
        weight = weight.squeeze(-1)
    lib_fallback("matmul", f"x {x.dtype} {tuple(x.shape)}, weight {weight.dtype} {tuple(weight.shape)}")
    return F.linear(x, _w(weight, x))


_LINEAR_EPILOGUE = True     # (tests switch it off to compare with the GEMM + element-wise pass pair)


def linear(x, weight, bias=None, act_slope=None, dropout_p=0.0, training=False):
    """bf16: ONE HIP GEMM with bias / LeakyReLU / dropout in its epilogue (_LinearEpiFn). Otherwise GEMM + one hand-written epilogue pass."""
    if act_slope is None and not (training and dropout_p > 0):
        if bias is None:
            return matmul_nt(x, weight)
        if not _gemm_ok(x, weight):
            if _gemm_f32_ok(x, weight) or _gemm_f32_widen_ok(x, weight):
                return matmul_nt(x, weight) + _w(bias, x)
            lib_fallback("matmul", f"x {x.dtype} {tuple(x.shape)}, weight {weight.dtype} {tuple(weight.shape)}")
            return F.linear(x, _w(weight, x), _w(bias, x))
    if _gemm_ok(x, weight) and _LINEAR_EPILOGUE:
        p = float(dropout_p) if training else 0.0
        return _LinearEpiFn.apply(x, weight, bias, -1.0 if act_slope is None else float(act_slope), p, next_seed() if p > 0 else 0)
    return bias_act_dropout(matmul_nt(x, weight), bias, act_slope, dropout_p, training)


_LSTM_WS = []   # sync blocks of the most recent persistent-recurrence launches (error words: include/tsasr_hip.h, tsasr_lstm_seq_workspace_bytes)


def _note_lstm_ws(ws):
    _LSTM_WS.append(ws[:256])
    del _LSTM_WS[:-4]


def lstm_timeouts():
    """Number of raised error words in the kept sync blocks (a host read): > 0 means a persistent LSTM launch gave up waiting for
    another workgroup - its outputs were poisoned with NaN, the optimizer skipped that step; the caller should raise."""
    n = 0
    for blk in _LSTM_WS:
        n += int((blk.view(torch.int32)[1::2] != 0).sum().item())
    return n


class _LstmFn(torch.autograd.Function):
    """Single-layer LSTM over [B,U,I] in bf16: input projection = HIP GEMM (the 28 embedding columns padded to 32, bias folded into the
    padding), recurrence = one persistent HIP kernel per direction (H in {256, 512}; else per-step HIP GEMM + cell kernel); the
    weight gradients are single HIP GEMMs over all (b,t). Every launch is graph-capturable."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, onehot_blank=None):
        H = w_hh.shape[1]
        dev = x.device
        if onehot_blank is not None:
            # x = token ids [B,U] of a frozen one-hot Embedding: the input projection is a column of W_ih per token - one launch writes
            # the gate pre-activations (fp32 weights and biases, no rounding to bf16) and the padded one-hot rows the backward contracts with
            B, U = x.shape
            I = w_ih.shape[1]
            Ip = (I + 2 + 7) // 8 * 8
            tok = x if x.dtype == torch.int64 else x.long()
            xp = torch.empty(B * U, Ip, dtype=torch.bfloat16, device=dev)
            gates = torch.empty(B, U, H, 4, dtype=torch.float32, device=dev)
            C.check(C.lib().tsasr_lstm_onehot_gates(C.ptr(tok.contiguous()), C.ptr(_f32(w_ih).contiguous()), C.ptr(_f32(b_ih).contiguous()),
                                                    C.ptr(_f32(b_hh).contiguous()), C.ptr(gates), C.ptr(xp), B, U, H, I, Ip, int(onehot_blank),
                                                    C.stream_ptr()), "tsasr_lstm_onehot_gates")
            return _LstmFn._recurrence(ctx, xp, gates.view(B * U, 4 * H), w_ih, w_hh, b_ih, b_hh, (B, U, I))
        B, U, I = x.shape
        # x-part of the gate pre-activations, laid out [B,U,H,4] (gate-minor: the kernels move a unit's four gates as one float4)
        perm = lambda t: t.view(4, H, *t.shape[1:]).transpose(0, 1).reshape(t.shape)  # noqa: E731
        # input projection on the HIP GEMM too: the inner dimension (28 embedding columns) is padded to a multiple of 8 and the bias
        # rides in two of the padding columns (x = 1 there; the weight columns hold the bias split into a bf16 high and low part,
        # i.e. ~16 mantissa bits, accumulated in fp32 by the MFMA) - no library GEMM, no fp32 copies of x and W_ih
        Ip = (I + 2 + 7) // 8 * 8
        xp = torch.zeros(B * U, Ip, dtype=torch.bfloat16, device=dev)
        xp[:, :I] = x.reshape(B * U, I)
        xp[:, I:I + 2] = 1.0
        bias = perm((b_ih + b_hh).float())
        b_hi = bias.to(torch.bfloat16)
        wp = torch.zeros(4 * H, Ip, dtype=torch.bfloat16, device=dev)
        wp[:, :I] = perm(w_ih)
        wp[:, I] = b_hi
        wp[:, I + 1] = bias - b_hi.float()
        gates = gemm_bf16(xp, wp, B * U, 4 * H, Ip, Ip, Ip, 0, 0, out_dtype=torch.float32)
        return _LstmFn._recurrence(ctx, xp, gates, w_ih, w_hh, b_ih, b_hh, (B, U, I))

    @staticmethod
    def _recurrence(ctx, xp, gates, w_ih, w_hh, b_ih, b_hh, in_shape):
        B, U, I = in_shape
        H = w_hh.shape[1]
        dev = xp.device
        c = torch.empty(B, U, H, dtype=torch.float32, device=dev)
        h = torch.empty(B, U, H, dtype=torch.bfloat16, device=dev)
        whh16 = _bf16_weight(w_hh).contiguous()
        lib, st = C.lib(), C.stream_ptr()
        ws = _ws(lib.tsasr_lstm_seq_workspace_bytes(B, U, H), dev)
        if lib.tsasr_lstm_seq_persistent(B, H, C.BF16):
            _note_lstm_ws(ws)
        with prof.region("lstm_fwd"):   # the whole recurrence: one persistent launch (H in {256, 512}), else one fused launch per step
            C.check(lib.tsasr_lstm_seq_fwd(C.ptr(gates), C.ptr(c), C.ptr(h), C.ptr(whh16), B, U, H, C.BF16, C.ptr(ws), ws.numel(), st),
                    "tsasr_lstm_seq_fwd")
        ctx.save_for_backward(xp, gates, c, h, whh16)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.in_shape = (B, U, I)
        return h

    @staticmethod
    def backward(ctx, dout):
        xp, gates, c, h, whh16 = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        B, U, I = ctx.in_shape
        Ip = xp.shape[1]
        H = w_hh.shape[1]
        dev = xp.device
        dout = dout.contiguous()
        dgates = torch.empty(B, U, 4 * H, dtype=torch.bfloat16, device=dev)
        whhT = _bf16_weight_t(w_hh)
        if whhT is None:
            whhT = whh16.t().contiguous()
        lib, st = C.lib(), C.stream_ptr()
        ws = _ws(lib.tsasr_lstm_seq_workspace_bytes(B, U, H), dev)
        if lib.tsasr_lstm_seq_persistent(B, H, C.BF16):
            _note_lstm_ws(ws)
        with prof.region("lstm_bwd"):   # dh = dout[:, t] + dgates[:, t+1] . W_hh, cell backward, t = U-1 .. 0 in one launch
            C.check(lib.tsasr_lstm_seq_bwd(C.ptr(gates), C.ptr(c), C.ptr(dout), C.ptr(dgates), C.ptr(whhT), B, U, H, C.BF16,
                                           C.ptr(ws), ws.numel(), st), "tsasr_lstm_seq_bwd")
        dg2 = dgates.view(B * U, 4 * H)
        h_prev = torch.zeros_like(h)
        h_prev[:, 1:] = h[:, :-1]
        dw_hh = gemm_bf16(dg2, h_prev.view(B * U, H), 4 * H, H, B * U, 4 * H, H, 1, 1, out_dtype=torch.float32)
        # dW_ih and the bias gradient in one GEMM against the padded input (its ones-column sums the gate gradients over b, t)
        dwp = gemm_bf16(dg2, xp, 4 * H, Ip, B * U, 4 * H, Ip, 1, 1, out_dtype=torch.float32)
        dw_ih, db = dwp[:, :I], dwp[:, I]
        dx = None
        if ctx.needs_input_grad[0]:
            wq = torch.zeros(4 * H, Ip, dtype=torch.bfloat16, device=dev)
            wq[:, :I] = w_ih
            dx = gemm_bf16(dg2, wq, B * U, Ip, 4 * H, 4 * H, Ip, 0, 1)[:, :I].reshape(B, U, I)
        return dx, dw_ih.to(w_ih.dtype), dw_hh.to(w_hh.dtype), db.to(b_ih.dtype), db.to(b_hh.dtype), None


def lstm_onehot_supported(tokens, rnn, num_embeddings):
    """True when ops.lstm_onehot takes this predictor: token ids on the GPU, single-layer unidirectional LSTM over V-1 one-hot inputs,
    more than one step, hidden size the recurrence kernels take."""
    H = rnn.hidden_size
    return (tokens.is_cuda and not tokens.dtype.is_floating_point and tokens.dim() == 2 and tokens.shape[1] > 1 and rnn.num_layers == 1
            and not rnn.bidirectional and rnn.bias and H % 16 == 0 and rnn.input_size == num_embeddings - 1 and rnn.input_size + 2 <= H)


def lstm_onehot(tokens, rnn, blank):
    """LSTM over one-hot embedded ``tokens`` [B,U] (frozen Embedding(consider_as_one_hot=True)): bf16 outputs [B,U,H]."""
    return _LstmFn.apply(tokens, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, int(blank))


class _LstmF32Fn(torch.autograd.Function):
    """Single-layer LSTM in exact fp32 arithmetic with an optional initial state (csrc/lstm_f32.hip): returns (out [B,U,H], hn, cn [B,H]).
    The recurrence is one HIP launch each way; dx / dW_ih / dW_hh / db are fp32 GEMMs over all (b, t) (csrc/gemm_f32.hip)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0, grad_mode=True):
        C.require_gpu(x, w_ih, w_hh)
        xc = x.contiguous()
        B, U, I = xc.shape
        H = w_hh.shape[1]
        f = lambda t: None if t is None else _f32(t).contiguous()  # noqa: E731
        wi, wh, bi, bh, h0c, c0c = f(w_ih), f(w_hh), f(b_ih), f(b_hh), f(h0), f(c0)
        # grad_mode = torch.is_grad_enabled() AT THE CALL (inside forward() autograd has it switched off, and needs_input_grad only mirrors
        # requires_grad): the searchers call this under no_grad - no gate / cell-state planes then
        train = bool(grad_mode) and any(ctx.needs_input_grad)
        hs = torch.empty(B, U, H, dtype=torch.float32, device=x.device)
        hn, cn = torch.empty(B, H, dtype=torch.float32, device=x.device), torch.empty(B, H, dtype=torch.float32, device=x.device)
        cs = torch.empty(B, U, H, dtype=torch.float32, device=x.device) if train else None
        gates = torch.empty(B, U, 4 * H, dtype=torch.float32, device=x.device) if train else None
        with prof.region("lstm_f32_fwd"):
            C.check(C.lib().tsasr_lstm_f32_fwd(C.ptr(xc), C.ptr(wi), C.ptr(wh), C.ptr(bi), C.ptr(bh), C.ptr(h0c), C.ptr(c0c), C.ptr(hs), C.ptr(cs),
                                               C.ptr(gates), C.ptr(hn), C.ptr(cn), B, U, I, H, C.stream_ptr()), "tsasr_lstm_f32_fwd")
        if train:
            ctx.save_for_backward(xc, wi, wh, hs, cs, gates, h0c, c0c)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        return hs, hn, cn

    @staticmethod
    def backward(ctx, dout, dhn, dcn):
        xc, wi, wh, hs, cs, gates, h0c, c0c = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        B, U, I = xc.shape
        H = wh.shape[1]
        f = lambda t: None if t is None else t.float().contiguous()  # noqa: E731
        dgates = torch.empty(B, U, 4 * H, dtype=torch.float32, device=xc.device)
        dh0 = torch.empty(B, H, dtype=torch.float32, device=xc.device) if ctx.needs_input_grad[5] else None
        dc0 = torch.empty(B, H, dtype=torch.float32, device=xc.device) if ctx.needs_input_grad[6] else None
        with prof.region("lstm_f32_bwd"):
            C.check(C.lib().tsasr_lstm_f32_bwd(C.ptr(f(dout)), C.ptr(f(dhn)), C.ptr(f(dcn)), C.ptr(gates), C.ptr(cs), C.ptr(c0c), C.ptr(wh),
                                               C.ptr(dgates), C.ptr(dh0), C.ptr(dc0), B, U, H, C.stream_ptr()), "tsasr_lstm_f32_bwd")
        dg2 = dgates.view(B * U, 4 * H)
        h_prev = torch.empty_like(hs)
        h_prev[:, 1:] = hs[:, :-1]
        h_prev[:, 0] = 0.0 if h0c is None else h0c
        G = 4 * H
        dx = gemm_f32(dg2, wi, B * U, I, G, G, I, 0, 1).view(B, U, I) if ctx.needs_input_grad[0] else None
        dw_ih = gemm_f32(dg2, xc.view(B * U, I), G, I, B * U, G, I, 1, 1) if ctx.needs_input_grad[1] else None
        dw_hh = gemm_f32(dg2, h_prev.view(B * U, H), G, H, B * U, G, H, 1, 1) if ctx.needs_input_grad[2] else None
        db = dg2.sum(0) if (b_ih is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])) else None
        return (dx, dw_ih, dw_hh, db if ctx.needs_input_grad[3] else None, db if (b_hh is not None and ctx.needs_input_grad[4]) else None, dh0, dc0, None)


def lstm_f32_ok(rnn):
    return rnn.num_layers == 1 and not rnn.bidirectional and rnn.hidden_size <= 1024 and rnn.batch_first and getattr(rnn, "proj_size", 0) == 0


def lstm(x, rnn, hx=None):
    """bf16 training path (no initial state): persistent HIP recurrence on MFMA (_LstmFn). Everything else the recipes do with the predictor -
    fp32 parity runs, the searchers' step-wise calls with a carried (h, c) - runs the exact-fp32 HIP kernels of csrc/lstm_f32.hip
    (_LstmF32Fn); only LSTM shapes neither kernel takes (several layers, bidirectional, H > 1024) fall back to the library module."""
    if (hx is None and x.dtype == torch.bfloat16 and rnn.num_layers == 1 and not rnn.bidirectional and rnn.hidden_size % 16 == 0
            and x.shape[1] > 1):
        out = _LstmFn.apply(x, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)
        return out, None
    if LSTM_F32_HIP and lstm_f32_ok(rnn) and x.is_cuda:
        h0, c0 = (None, None) if hx is None else (hx[0][0], hx[1][0])       # torch's (num_layers, B, H)
        b_ih, b_hh = (rnn.bias_ih_l0, rnn.bias_hh_l0) if rnn.bias else (None, None)
        out, hn, cn = _LstmF32Fn.apply(x.float(), rnn.weight_ih_l0, rnn.weight_hh_l0, b_ih, b_hh, h0, c0, torch.is_grad_enabled())
        return out.to(x.dtype), (hn.unsqueeze(0), cn.unsqueeze(0))
    lib_fallback("lstm", f"num_layers {rnn.num_layers}, bidirectional {rnn.bidirectional}, hidden {rnn.hidden_size}")
    out, hn = rnn(x.float(), hx) if hx is not None else rnn(x.float())
    return out.to(x.dtype), hn


LSTM_F32_HIP = True       # (tests compare with the library module by switching it off)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, slope):
        C.require_gpu(x, weight, bias)
        D = weight.numel()
        xc = x.contiguous()
        M = xc.numel() // D
        g, b = _f32(weight).reshape(-1).contiguous(), _f32(bias).reshape(-1).contiguous()
        y = torch.empty_like(xc)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        with prof.region("layernorm_fwd"):
            C.check(C.lib().tsasr_layernorm_fwd(C.ptr(xc), C.ptr(g), C.ptr(b), C.ptr(y), C.ptr(mean), C.ptr(rstd), M, D, float(eps),
                                                float(slope), C.io_dtype(xc), C.stream_ptr()), "tsasr_layernorm_fwd")
        ctx.save_for_backward(xc, g, b, mean, rstd)
        ctx.slope, ctx.params = float(slope), (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, g, b, mean, rstd = ctx.saved_tensors
        D = g.numel()
        M = xc.numel() // D
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        dg, db = torch.empty_like(g), torch.empty_like(b)
        _keep(dg, db)
        ws = _ws(C.lib().tsasr_layernorm_bwd_workspace_bytes(M, D), xc.device)
        with prof.region("layernorm_bwd"):
            C.check(C.lib().tsasr_layernorm_bwd(C.ptr(dy), C.ptr(xc), C.ptr(g), C.ptr(b), C.ptr(mean), C.ptr(rstd), C.ptr(dx), C.ptr(dg),
                                                C.ptr(db), M, D, ctx.slope, C.io_dtype(xc), C.ptr(ws), ws.numel(), C.stream_ptr()),
                    "tsasr_layernorm_bwd")
        return dx, _pgrad(ctx.params[0], dg), _pgrad(ctx.params[1], db), None, None


class _LayerNormResFn(torch.autograd.Function):
    """(LayerNorm(x), x): the second output is x itself, handed on to whoever reads it as a residual. In backward the two gradients
    arrive together and the LayerNorm backward kernel sums them (no separate add kernel, no extra 4 MB round trip per layer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C.require_gpu(x, weight, bias)
        ctx.set_materialize_grads(False)
        D = weight.numel()
        xc = x.contiguous()
        M = xc.numel() // D
        g, b = _f32(weight).reshape(-1).contiguous(), _f32(bias).reshape(-1).contiguous()
        y = torch.empty_like(xc)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        with prof.region("layernorm_fwd"):
            C.check(C.lib().tsasr_layernorm_fwd(C.ptr(xc), C.ptr(g), C.ptr(b), C.ptr(y), C.ptr(mean), C.ptr(rstd), M, D, float(eps),
                                                -1.0, C.io_dtype(xc), C.stream_ptr()), "tsasr_layernorm_fwd")
        ctx.save_for_backward(xc, g, b, mean, rstd)
        ctx.params = (weight, bias)
        return y, xc.view_as(xc)

    @staticmethod
    def backward(ctx, dy, dres):
        xc, g, b, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        D = g.numel()
        M = xc.numel() // D
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        dg, db = torch.empty_like(g), torch.empty_like(b)
        _keep(dg, db)
        ws = _ws(C.lib().tsasr_layernorm_bwd_workspace_bytes(M, D), xc.device)
        with prof.region("layernorm_bwd"):
            if dres is None:
                C.check(C.lib().tsasr_layernorm_bwd(C.ptr(dy), C.ptr(xc), C.ptr(g), C.ptr(b), C.ptr(mean), C.ptr(rstd), C.ptr(dx), C.ptr(dg),
                                                    C.ptr(db), M, D, -1.0, C.io_dtype(xc), C.ptr(ws), ws.numel(), C.stream_ptr()),
                        "tsasr_layernorm_bwd")
            else:
                C.check(C.lib().tsasr_layernorm_bwd_add(C.ptr(dy), C.ptr(dres.contiguous()), C.ptr(xc), C.ptr(g), C.ptr(b), C.ptr(mean), C.ptr(rstd),
                                                        C.ptr(dx), C.ptr(dg), C.ptr(db), M, D, -1.0, C.io_dtype(xc), C.ptr(ws), ws.numel(),
                                                        C.stream_ptr()), "tsasr_layernorm_bwd_add")
        return dx, _pgrad(ctx.params[0], dg), _pgrad(ctx.params[1], db), None


def layer_norm_res(x, weight, bias, eps):
    """(LayerNorm(x), x) for an x that is also used as a residual afterwards (use the returned alias for that)."""
    return _LayerNormResFn.apply(x, weight, bias, eps)


def layer_norm(x, weight, bias, eps, act_slope=None):
    """LayerNorm over the trailing dims covered by ``weight`` (1-D model width or the front-end's [F, C]),
    optionally fused with the LeakyReLU that follows it."""
    return _LayerNormFn.apply(x, weight, bias, eps, -1.0 if act_slope is None else act_slope)


def layer_norm2(x, weight, bias, eps, act_slope=None):
    return layer_norm(x, weight, bias, eps, act_slope)


class _BiasActDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, p, seed):
        C.require_gpu(x)
        xc = x.contiguous()
        N = xc.shape[-1]
        M = xc.numel() // N
        b = None if bias is None else _f32(bias).contiguous()
        y = torch.empty_like(xc)
        with prof.region("bias_act_dropout_fwd"):
            C.check(C.lib().tsasr_bias_act_dropout_fwd(C.ptr(xc), C.ptr(b), C.ptr(y), M, N, float(slope), float(p), seed,
                                                       C.ptr(seed_state(xc.device)), C.io_dtype(xc), C.stream_ptr()), "tsasr_bias_act_dropout_fwd")
        ctx.save_for_backward(y)
        ctx.cfg = (float(slope), float(p), seed, bias is not None, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        slope, p, seed, has_bias, bias_param = ctx.cfg
        N = y.shape[-1]
        M = y.numel() // N
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        db = torch.empty(N, dtype=torch.float32, device=y.device) if has_bias else None
        _keep(db)
        ws = _ws(C.lib().tsasr_colpart_workspace_bytes(M, N), y.device) if has_bias else None
        with prof.region("bias_act_dropout_bwd"):
            C.check(C.lib().tsasr_bias_act_dropout_bwd(C.ptr(dy), C.ptr(y), C.ptr(dx), C.ptr(db), M, N, slope, p, seed,
                                                       C.ptr(seed_state(y.device)), C.io_dtype(y),
                                                       C.ptr(ws), 0 if ws is None else ws.numel(), C.stream_ptr()),
                    "tsasr_bias_act_dropout_bwd")
        return dx, (_pgrad(bias_param, db) if has_bias else None), None, None, None


def bias_act_dropout(x, bias, act_slope, p, training):
    """dropout_p(LeakyReLU(x + bias)) in one pass (act_slope None = no activation)."""
    p = float(p) if training else 0.0
    return _BiasActDropoutFn.apply(x, bias, -1.0 if act_slope is None else act_slope, p, next_seed() if p > 0 else 0)


class _DropoutAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, res, alpha, p, seed, valid_lens, trows, p2=0.0, seed2=0):
        C.require_gpu(x)
        xc = x.contiguous()
        N = xc.shape[-1]
        M = xc.numel() // N
        b = None if bias is None else _f32(bias).contiguous()
        r = None
        if res is not None:
            r = res.expand_as(xc).contiguous() if res.shape != xc.shape else res.contiguous()
        out = torch.empty_like(xc)
        with prof.region("dropout_add_fwd"):
            C.check(C.lib().tsasr_dropout_add2_fwd(C.ptr(xc), C.ptr(b), C.ptr(r), C.ptr(out), M, N, float(alpha), float(p), seed, float(p2), seed2,
                                                   C.ptr(seed_state(xc.device)), C.ptr(valid_lens), int(trows), C.io_dtype(xc), C.stream_ptr()),
                    "tsasr_dropout_add2_fwd")
        ctx.save_for_backward(valid_lens)
        ctx.cfg = (float(alpha), float(p), seed, int(trows), bias is not None, bias,
                   None if res is None else res.shape, xc.shape, float(p2), seed2)
        return out

    @staticmethod
    def backward(ctx, dout):
        (valid_lens,) = ctx.saved_tensors
        alpha, p, seed, trows, has_bias, bias_param, rshape, xshape, p2, seed2 = ctx.cfg
        dout = dout.contiguous()
        N = xshape[-1]
        M = dout.numel() // N
        dx = torch.empty_like(dout)
        g2 = torch.empty_like(dout) if p2 > 0 else None      # gradient behind the outer dropout (= gradient of the residual input)
        db = torch.empty(N, dtype=torch.float32, device=dout.device) if has_bias else None
        _keep(db)
        ws = _ws(C.lib().tsasr_colpart_workspace_bytes(M, N), dout.device) if has_bias else None
        with prof.region("dropout_add_bwd"):
            C.check(C.lib().tsasr_dropout_add2_bwd(C.ptr(dout), C.ptr(dx), C.ptr(g2), C.ptr(db), M, N, alpha, p, seed, p2, seed2,
                                                   C.ptr(seed_state(dout.device)), C.ptr(valid_lens), trows,
                                                   C.io_dtype(dout), C.ptr(ws), 0 if ws is None else ws.numel(), C.stream_ptr()),
                    "tsasr_dropout_add2_bwd")
        dres = None
        if rshape is not None:
            gsrc = g2 if g2 is not None else dout
            dres = gsrc if tuple(rshape) == tuple(xshape) else gsrc.sum_to_size(rshape)
        return dx, (_pgrad(bias_param, db) if has_bias else None), dres, None, None, None, None, None, None, None


def dropout_add(x, bias=None, res=None, alpha=1.0, p=0.0, training=False, valid_lens=None, outer_p=0.0):
    """dropout_outer_p( res + alpha * timemask(dropout_p(x + bias)) ); x is [B, T, N] when valid_lens (int32 [B]) is given. The outer
    dropout (a front-end ConvBlock's last Dropout) rides in the same pass."""
    p = float(p) if training else 0.0
    p2 = float(outer_p) if training else 0.0
    trows = x.shape[-2] if valid_lens is not None else 0
    return _DropoutAddFn.apply(x, bias, res, alpha, p, next_seed() if p > 0 else 0, valid_lens, trows, p2, next_seed() if p2 > 0 else 0)


class _AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (res + alpha*timemask(dropout(x + bias)), LayerNorm(s)) - the seam between two Conformer sub-blocks in one pass;
    the backward folds the residual-path gradient, the LayerNorm backward and the dropout/mask backward into one kernel."""

    @staticmethod
    def forward(ctx, x, bias, res, gamma, beta, alpha, p, seed, valid_lens, trows, eps):
        C.require_gpu(x, res, gamma, beta)
        ctx.set_materialize_grads(False)      # an unused output (the last seam of a layer drops s) must arrive as None, not as a
        xc, r = x.contiguous(), res.contiguous()   # freshly zero-filled 4 MB tensor that the backward kernel would then read
        D = xc.shape[-1]
        M = xc.numel() // D
        b = None if bias is None else _f32(bias).contiguous()
        g, bt = _f32(gamma).contiguous(), _f32(beta).contiguous()
        s_out, y = torch.empty_like(xc), torch.empty_like(xc)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        with prof.region("add_layernorm_fwd"):
            C.check(C.lib().tsasr_add_layernorm_fwd(C.ptr(xc), C.ptr(b), C.ptr(r), C.ptr(s_out), C.ptr(y), C.ptr(mean), C.ptr(rstd),
                                                    C.ptr(g), C.ptr(bt), M, D, float(alpha), float(p), seed, C.ptr(seed_state(xc.device)),
                                                    C.ptr(valid_lens), int(trows), float(eps), C.io_dtype(xc), C.stream_ptr()),
                    "tsasr_add_layernorm_fwd")
        ctx.save_for_backward(s_out, g, mean, rstd, valid_lens)
        ctx.cfg = (float(alpha), float(p), seed, int(trows), bias, gamma, beta)
        return s_out, y

    @staticmethod
    def backward(ctx, ds_in, dy):
        s_out, g, mean, rstd, valid_lens = ctx.saved_tensors
        alpha, p, seed, trows, bias_param, gamma, beta = ctx.cfg
        D = s_out.shape[-1]
        M = s_out.numel() // D
        if dy is None:                                        # y unused downstream: only the residual tail has a gradient
            dy = torch.zeros_like(s_out)
        dy = dy.contiguous()
        ds_in = None if ds_in is None else ds_in.contiguous()
        dres, dx = torch.empty_like(s_out), torch.empty_like(s_out)
        dg = torch.empty(D, dtype=torch.float32, device=s_out.device)
        dbt = torch.empty_like(dg)
        db = torch.empty_like(dg) if bias_param is not None else None
        _keep(dg, dbt, db)
        ws = _ws(C.lib().tsasr_add_layernorm_bwd_workspace_bytes(M, D), s_out.device)
        with prof.region("add_layernorm_bwd"):
            C.check(C.lib().tsasr_add_layernorm_bwd(C.ptr(dy), C.ptr(ds_in), C.ptr(s_out), C.ptr(g), C.ptr(mean), C.ptr(rstd), C.ptr(dres),
                                                    C.ptr(dx), C.ptr(dg), C.ptr(dbt), C.ptr(db), M, D, alpha, p, seed,
                                                    C.ptr(seed_state(s_out.device)), C.ptr(valid_lens), trows, C.io_dtype(s_out),
                                                    C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_add_layernorm_bwd")
        return (dx, (_pgrad(bias_param, db) if bias_param is not None else None), dres, _pgrad(gamma, dg), _pgrad(beta, dbt),
                None, None, None, None, None, None)


def add_layer_norm(x, bias, res, ln, alpha=1.0, p=0.0, training=False, valid_lens=None, eps=1e-5):
    """Returns (s, LayerNorm(s)) with s = res + alpha * timemask(dropout_p(x + bias)); ``ln`` holds weight/bias [D]."""
    p = float(p) if training else 0.0
    trows = x.shape[-2] if valid_lens is not None else 0
    return _AddLayerNormFn.apply(x, bias, res, ln.weight, ln.bias, alpha, p, next_seed() if p > 0 else 0, valid_lens, trows, eps)


LINEAR_LN_FUSED = os.environ.get("TSASR_LINEAR_LN", "1") != "0"     # (tests / A-B: "0" = the GEMM and the row kernel as two launches)


class _LinearAddLayerNormFn(torch.autograd.Function):
    """(s, y) of add_layer_norm applied to x = xin . W^T without x ever being written: csrc/linear_ln.hip (bf16, N = K = 256: the attention's
    out_proj and the convolution module's last point-wise convolution). Same bits as matmul_nt + add_layer_norm; the backward is theirs."""

    @staticmethod
    def forward(ctx, xin, weight, bias, res, gamma, beta, alpha, p, seed, valid_lens, trows, eps):
        C.require_gpu(xin, res, gamma, beta)
        ctx.set_materialize_grads(False)
        N, K = weight.shape[0], weight.shape[1]
        x2 = xin.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w16 = _bf16_weight(weight).view(N, K) if weight.dim() == 3 else _bf16_weight(weight)
        if w16.stride(1) != 1 or w16.stride(0) % 8 != 0 or w16.data_ptr() % 16 != 0:
            w16 = w16.contiguous()
        r = res.contiguous()
        M = x2.shape[0]
        b = None if bias is None else _f32(bias).contiguous()
        g, bt = _f32(gamma).contiguous(), _f32(beta).contiguous()
        s_out, y = torch.empty_like(r), torch.empty_like(r)
        mean = torch.empty(M, dtype=torch.float32, device=r.device)
        rstd = torch.empty(M, dtype=torch.float32, device=r.device)
        with prof.region("linear_add_layernorm_fwd"):
            C.check(C.lib().tsasr_linear_add_layernorm_fwd(C.ptr(x2), x2.stride(0), C.ptr(w16), w16.stride(0), C.ptr(b), C.ptr(r), C.ptr(s_out),
                                                           C.ptr(y), C.ptr(mean), C.ptr(rstd), C.ptr(g), C.ptr(bt), M, N, K, float(alpha), float(p),
                                                           seed, C.ptr(seed_state(r.device)), C.ptr(valid_lens), int(trows), float(eps),
                                                           C.stream_ptr()), "tsasr_linear_add_layernorm_fwd")
        ctx.save_for_backward(x2, w16, s_out, g, mean, rstd, valid_lens)
        ctx.cfg = (float(alpha), float(p), seed, int(trows), bias, gamma, beta, weight, xin.shape)
        return s_out, y

    @staticmethod
    def backward(ctx, ds_in, dy):
        x2, w16, s_out, g, mean, rstd, valid_lens = ctx.saved_tensors
        alpha, p, seed, trows, bias_param, gamma, beta, weight, xshape = ctx.cfg
        D = s_out.shape[-1]
        M = s_out.numel() // D
        if dy is None:
            dy = torch.zeros_like(s_out)
        dy = dy.contiguous()
        ds_in = None if ds_in is None else ds_in.contiguous()
        dres, dx = torch.empty_like(s_out), torch.empty_like(s_out)
        dg = torch.empty(D, dtype=torch.float32, device=s_out.device)
        dbt = torch.empty_like(dg)
        db = torch.empty_like(dg) if bias_param is not None else None
        _keep(dg, dbt, db)
        ws = _ws(C.lib().tsasr_add_layernorm_bwd_workspace_bytes(M, D), s_out.device)
        with prof.region("add_layernorm_bwd"):
            C.check(C.lib().tsasr_add_layernorm_bwd(C.ptr(dy), C.ptr(ds_in), C.ptr(s_out), C.ptr(g), C.ptr(mean), C.ptr(rstd), C.ptr(dres),
                                                    C.ptr(dx), C.ptr(dg), C.ptr(dbt), C.ptr(db), M, D, alpha, p, seed,
                                                    C.ptr(seed_state(s_out.device)), C.ptr(valid_lens), trows, C.io_dtype(s_out),
                                                    C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_add_layernorm_bwd")
        N, K = weight.shape[0], weight.shape[1]
        dx2 = dx.view(M, N)
        dxin = _dgrad(dx2, weight, w16, M, N, K).view(xshape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            sink = _GRAD_SINK
            if (sink is not None and weight.is_leaf and sink.accepts(weight) and weight.grad.dtype == torch.float32
                    and weight.grad.is_contiguous()):
                _wgrad_into(sink, weight, weight.grad.view(N, K), dx2, x2)
            else:
                dw = gemm_bf16(dx2, x2, N, K, M, N, K, 1, 1, out_dtype=torch.float32).to(weight.dtype).view(weight.shape)
        return (dxin, dw, (_pgrad(bias_param, db) if bias_param is not None else None), dres, _pgrad(gamma, dg), _pgrad(beta, dbt),
                None, None, None, None, None, None)


def linear_add_layer_norm(xin, weight, bias, res, ln, alpha=1.0, p=0.0, training=False, valid_lens=None, eps=1e-5):
    """add_layer_norm(matmul_nt(xin, weight), bias, res, ln, ...): one launch where csrc/linear_ln.hip has the shape, else the pair."""
    if (LINEAR_LN_FUSED and _gemm_ok(xin, weight) and res.dtype == torch.bfloat16 and weight.shape[0] == 256 and weight.shape[1] == 256
            and res.shape[-1] == 256 and xin.shape[:-1] == res.shape[:-1]):
        p = float(p) if training else 0.0
        trows = res.shape[-2] if valid_lens is not None else 0
        return _LinearAddLayerNormFn.apply(xin, weight, bias, res, ln.weight, ln.bias, alpha, p, next_seed() if p > 0 else 0, valid_lens, trows, eps)
    return add_layer_norm(matmul_nt(xin, weight), bias, res, ln, alpha, p, training, valid_lens, eps)


class _AddLayerNorm2Fn(torch.autograd.Function):
    """(y, z) = (LN(s; g1, b1), LN(y; g2, b2)), s = res + alpha*timemask(dropout(x + bias)): _AddLayerNormFn followed by _LayerNormResFn
    in one launch each way (norm2 of a Conformer layer + the next layer's first LayerNorm, or the encoder's final norm), same bits."""

    @staticmethod
    def forward(ctx, x, bias, res, g1, b1, g2, b2, alpha, p, seed, valid_lens, trows, eps, eps2):
        C.require_gpu(x, res, g1, b1, g2, b2)
        ctx.set_materialize_grads(False)
        xc, r = x.contiguous(), res.contiguous()
        D = xc.shape[-1]
        M = xc.numel() // D
        b = None if bias is None else _f32(bias).contiguous()
        ga, ba, gb, bb = (_f32(t).reshape(-1).contiguous() for t in (g1, b1, g2, b2))
        s_out, y, z = torch.empty_like(xc), torch.empty_like(xc), torch.empty_like(xc)
        stats = torch.empty(4, M, dtype=torch.float32, device=x.device)     # mean, rstd of s ; mean, rstd of y
        with prof.region("add_layernorm2_fwd"):
            C.check(C.lib().tsasr_add_layernorm2_fwd(C.ptr(xc), C.ptr(b), C.ptr(r), C.ptr(s_out), C.ptr(y), C.ptr(z), C.ptr(stats[0]),
                                                     C.ptr(stats[1]), C.ptr(stats[2]), C.ptr(stats[3]), C.ptr(ga), C.ptr(ba), C.ptr(gb), C.ptr(bb),
                                                     M, D, float(alpha), float(p), seed, C.ptr(seed_state(xc.device)), C.ptr(valid_lens),
                                                     int(trows), float(eps), float(eps2), C.io_dtype(xc), C.stream_ptr()), "tsasr_add_layernorm2_fwd")
        ctx.save_for_backward(s_out, ga, ba, gb, stats, valid_lens)
        ctx.cfg = (float(alpha), float(p), seed, int(trows), bias, g1, b1, g2, b2)
        return y, z

    @staticmethod
    def backward(ctx, dy, dz):
        s_out, ga, ba, gb, stats, valid_lens = ctx.saved_tensors
        alpha, p, seed, trows, bias_param, g1, b1, g2, b2 = ctx.cfg
        D = s_out.shape[-1]
        M = s_out.numel() // D
        if dz is None:
            dz = torch.zeros_like(s_out)
        dz = dz.contiguous()
        dy = None if dy is None else dy.contiguous()
        dres, dx = torch.empty_like(s_out), torch.empty_like(s_out)
        dg1, dbt1, dg2, dbt2 = (torch.empty(D, dtype=torch.float32, device=s_out.device) for _ in range(4))
        db = torch.empty(D, dtype=torch.float32, device=s_out.device) if bias_param is not None else None
        _keep(dg1, dbt1, dg2, dbt2, db)
        ws = _ws(C.lib().tsasr_add_layernorm2_bwd_workspace_bytes(M, D), s_out.device)
        with prof.region("add_layernorm2_bwd"):
            C.check(C.lib().tsasr_add_layernorm2_bwd(C.ptr(dz), C.ptr(dy), None, C.ptr(s_out), C.ptr(ga), C.ptr(ba), C.ptr(gb), C.ptr(stats[0]),
                                                     C.ptr(stats[1]), C.ptr(stats[2]), C.ptr(stats[3]), C.ptr(dres), C.ptr(dx), C.ptr(dg1),
                                                     C.ptr(dbt1), C.ptr(db), C.ptr(dg2), C.ptr(dbt2), M, D, alpha, p, seed,
                                                     C.ptr(seed_state(s_out.device)), C.ptr(valid_lens), trows, C.io_dtype(s_out),
                                                     C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_add_layernorm2_bwd")
        return (dx, (_pgrad(bias_param, db) if bias_param is not None else None), dres, _pgrad(g1, dg1), _pgrad(b1, dbt1),
                _pgrad(g2, dg2), _pgrad(b2, dbt2), None, None, None, None, None, None, None)


_FUSED_LN_PAIR = True


def add_layer_norm2_supported(x):
    return _FUSED_LN_PAIR and x.is_cuda and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024


def add_layer_norm2(x, bias, res, ln, ln_next, alpha=1.0, p=0.0, training=False, valid_lens=None, eps=1e-5, eps2=1e-5):
    """Returns (y, z) = (ln(s), ln_next(ln(s))) with s = res + alpha * timemask(dropout_p(x + bias))."""
    p = float(p) if training else 0.0
    trows = x.shape[-2] if valid_lens is not None else 0
    return _AddLayerNorm2Fn.apply(x, bias, res, ln.weight, ln.bias, ln_next.weight, ln_next.bias, alpha, p, next_seed() if p > 0 else 0,
                                  valid_lens, trows, eps, eps2)


class _MeanPoolFn(torch.autograd.Function):
    """Masked mean over time of [B,T,D] -> [B,1,D] (train_librispeechmix_scratch.py:52-64), one HIP launch each way."""

    @staticmethod
    def forward(ctx, x, rel_lens):
        C.require_gpu(x, rel_lens)
        xc = x.contiguous()
        B, T, D = xc.shape
        rel = rel_lens.detach().float().contiguous()
        out = torch.empty(B, 1, D, dtype=xc.dtype, device=xc.device)
        C.check(C.lib().tsasr_mean_pool_fwd(C.ptr(xc), C.ptr(rel), C.ptr(out), B, T, D, C.io_dtype(xc), C.stream_ptr()), "tsasr_mean_pool_fwd")
        ctx.save_for_backward(rel)
        ctx.shape = (B, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (rel,) = ctx.saved_tensors
        B, T, D = ctx.shape
        dout = dout.contiguous()
        dx = torch.empty(B, T, D, dtype=dout.dtype, device=dout.device)
        C.check(C.lib().tsasr_mean_pool_bwd(C.ptr(dout), C.ptr(rel), C.ptr(dx), B, T, D, C.io_dtype(dout), C.stream_ptr()), "tsasr_mean_pool_bwd")
        return dx, None


def mean_pool(x, rel_lens):
    return _MeanPoolFn.apply(x, rel_lens)


class _InjectFn(torch.autograd.Function):
    """src [B,T,D] (+ | *) spk [B,1,D] - the `sum` / `prod` speaker-embedding injections (models/conformer.py:247-253) as one HIP launch
    each way (tsasr_inject_fwd / _bwd) instead of ATen's broadcast op + its sum-to-size backward."""

    @staticmethod
    def forward(ctx, src, spk, mode):
        C.require_gpu(src, spk)
        srcc, spkc = src.contiguous(), spk.contiguous()
        B, T, D = srcc.shape
        out = torch.empty_like(srcc)
        with prof.region("inject_fwd"):
            C.check(C.lib().tsasr_inject_fwd(C.ptr(srcc), C.ptr(spkc), C.ptr(out), B, T, D, mode, C.io_dtype(srcc), C.stream_ptr()), "tsasr_inject_fwd")
        ctx.save_for_backward(*((srcc, spkc) if mode == 1 else ()))
        ctx.mode, ctx.shape = mode, (B, T, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, D = ctx.shape
        srcc, spkc = ctx.saved_tensors if ctx.mode == 1 else (None, None)
        dout = dout.contiguous()
        dsrc = torch.empty_like(dout) if ctx.mode == 1 else None      # sum: the gradient of src is dout itself, nothing to write
        dspk = torch.empty(B, 1, D, dtype=dout.dtype, device=dout.device)
        with prof.region("inject_bwd"):
            C.check(C.lib().tsasr_inject_bwd(C.ptr(dout), C.ptr(srcc), C.ptr(spkc), C.ptr(dsrc), C.ptr(dspk), B, T, D, ctx.mode,
                                             C.io_dtype(dout), C.stream_ptr()), "tsasr_inject_bwd")
        return (dout if dsrc is None else dsrc), dspk, None


def inject_ok(src, spk):
    return (src.is_cuda and src.dim() == 3 and spk.dim() == 3 and spk.shape[1] == 1 and spk.shape[0] == src.shape[0] and spk.shape[2] == src.shape[2]
            and src.shape[2] % 8 == 0 and src.dtype == spk.dtype and src.dtype in (torch.float32, torch.bfloat16))


def inject(src, spk, mode):
    """mode "sum" | "prod" (see _InjectFn)."""
    return _InjectFn.apply(src, spk, 1 if mode == "prod" else 0)


def abs_lengths(rel, dim, mode=0):
    """int32 [B] absolute lengths from relative ones on the device, one launch: mode 0 = (rel * dim).round() (half to even, as
    torch.round: models/conformer.py:272, SB/nnet/losses.py:58-59), 1 = floor (SB/nnet/RNN.py:35), 2 = ceil clamped to dim."""
    import ctypes
    C.require_gpu(rel)
    # the recipe asks for the same lengths several times per step (encoder mask, joint, loss): one launch per (tensor, size, rule,
    # stream). An entry holds its source tensor, so the address cannot be handed to another tensor while the entry lives; an in-place
    # update of the source changes its version. Cleared at every begin_step.
    key = (rel.data_ptr(), rel._version, tuple(rel.shape), rel.dtype, int(dim), int(mode), torch.cuda.current_stream().cuda_stream)
    hit = _LEN_CACHE.get(key)
    if hit is not None:
        return hit[1]
    r = rel.detach().float().contiguous()
    out = torch.empty(r.shape[0], dtype=torch.int32, device=r.device)
    pr, po = (ctypes.c_void_p * 1)(r.data_ptr()), (ctypes.c_void_p * 1)(out.data_ptr())
    pd, pm = (ctypes.c_int * 1)(int(dim)), (ctypes.c_int * 1)(int(mode))
    C.check(C.lib().tsasr_abs_lengths(pr, po, pd, pm, 1, r.shape[0], C.stream_ptr()), "tsasr_abs_lengths")
    if len(_LEN_CACHE) >= 32:
        _LEN_CACHE.clear()
    _LEN_CACHE[key] = (rel, out)
    return out


def count_nonfinite(x, counter):
    """counter (int32 device scalar) += number of NaN / Inf in x (fp32, any shape): one launch."""
    C.require_gpu(x, counter)
    xf = x.detach().float().contiguous().view(-1)
    C.check(C.lib().tsasr_count_nonfinite(C.ptr(xf), xf.numel(), C.ptr(counter), C.stream_ptr()), "tsasr_count_nonfinite")


def mask_time(x, valid_lens):
    """Zero frames t >= valid_lens[b] of x [B,T,D] (ConvolutionModule's masked_fill_, Conformer.py:113-114)."""
    return dropout_add(x, None, None, 1.0, 0.0, False, valid_lens)


# ---------------------------------------------------------------------------------------------------------
def fbank(wav, window, fbank_matrix, n_fft, hop, win, top_db, amin, out_dtype=torch.float32):
    """wav [B,L] -> log-mel [B, 1+L//hop, n_mels]: HIP kernel (in-LDS FFT-512 + power + mel + dB + per-utterance floor).
    Features carry no gradient on this path (the reference's Fbank is frozen: requires_grad=False)."""
    C.require_gpu(wav)
    if n_fft != 512 or win != 512:
        raise NotImplementedError("ts-asr_amd.Fbank: n_fft = win_length = 512 samples (all TS-ASR YAMLs: n_fft 512, 32 ms at 16 kHz)")
    w = wav.detach().float().contiguous()
    B, L = w.shape
    T = 1 + L // hop
    n_mels = fbank_matrix.shape[1]
    out = torch.empty(B, T, n_mels, dtype=out_dtype, device=w.device)
    ws = _ws(C.lib().tsasr_fbank_workspace_bytes(B, T, n_mels), w.device)
    with prof.region("fbank"):
        C.check(C.lib().tsasr_fbank_fwd(C.ptr(w), C.ptr(window.float().contiguous()), C.ptr(fbank_matrix.float().contiguous()), C.ptr(out),
                                        B, L, T, n_mels, int(hop), float(top_db), float(amin), C.io_dtype(out), C.ptr(ws), ws.numel(),
                                        C.stream_ptr()), "tsasr_fbank_fwd")
    return out


def sentence_norm(x, abs_lens, eps, out_dtype=None):
    """Per-utterance mean / unbiased std over the first abs_lens[b] frames, applied to the whole padded row (HIP kernel)."""
    C.require_gpu(x, abs_lens)
    xc = x.detach().contiguous()
    B, T, Fq = xc.shape
    y = torch.empty(B, T, Fq, dtype=out_dtype or xc.dtype, device=xc.device)
    with prof.region("sentence_norm"):
        C.check(C.lib().tsasr_sentence_norm_fwd(C.ptr(xc), C.ptr(abs_lens.to(torch.int32).contiguous()), C.ptr(y), B, T, Fq, float(eps),
                                                C.io_dtype(xc), C.io_dtype(y), C.stream_ptr()), "tsasr_sentence_norm_fwd")
    return y


def spec_augment_draw(B, T, Fq, window, n_freq, f_range, n_time, t_range, device):
    """All random numbers of one SpecAugment call, drawn by a HIP kernel into a device int32 table (no host round trip; the
    stream id comes from ``next_seed`` + the device step counter, so a captured call draws afresh on every replay)."""
    lib = C.lib()
    params = torch.empty(lib.tsasr_specaug_params_words(B, n_freq, n_time), dtype=torch.int32, device=device)
    C.require_gpu(params)
    C.check(lib.tsasr_specaug_draw(C.ptr(params), B, T, Fq, int(window), n_freq, int(f_range[0]), int(f_range[1]), n_time,
                                   int(t_range[0]), int(t_range[1]), next_seed(), C.ptr(seed_state(params.device)), C.stream_ptr()),
            "tsasr_specaug_draw")
    return params


def spec_augment_apply(x, params, n_freq, n_time, replace_with_zero):
    """y = time masks(frequency masks(time warp(x))) for x [B,T,F] with the draws of ``params`` (layout: include/tsasr_hip.h)."""
    C.require_gpu(x, params)
    xc = x.detach().contiguous()
    B, T, Fq = xc.shape
    lib = C.lib()
    if params.numel() != lib.tsasr_specaug_params_words(B, n_freq, n_time) or params.dtype != torch.int32:
        raise ValueError("spec_augment_apply: params must be the int32 table of spec_augment_draw for this batch and mask counts")
    y = torch.empty_like(xc)
    ws = _ws(lib.tsasr_specaug_workspace_bytes(), xc.device)
    with prof.region("spec_augment"):
        C.check(lib.tsasr_specaug_apply(C.ptr(xc), C.ptr(y), C.ptr(params), B, T, Fq, n_freq, n_time, int(bool(replace_with_zero)),
                                        C.io_dtype(xc), C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_specaug_apply")
    return y


def resample_out_len(n_in, orig_freq, new_freq):
    return int(C.lib().tsasr_resample_out_len(int(n_in), int(orig_freq), int(new_freq)))


def resample(wav, weights, first, orig_freq, new_freq):
    """Polyphase resampling of wav [B,L] fp32 with the filter bank ``weights`` [P,W] / ``first`` [P] (device tensors)."""
    import math
    C.require_gpu(wav, weights, first)
    x = wav.detach().float().contiguous()
    B, L = x.shape
    base = math.gcd(orig_freq, new_freq)
    P, stride = new_freq // base, orig_freq // base
    n_out = resample_out_len(L, orig_freq, new_freq)
    y = torch.empty(B, n_out, dtype=torch.float32, device=x.device)
    with prof.region("resample"):
        C.check(C.lib().tsasr_resample_fwd(C.ptr(x), C.ptr(y), C.ptr(weights), C.ptr(first), B, L, n_out, P, stride, weights.shape[1],
                                           C.stream_ptr()), "tsasr_resample_fwd")
    return y


def colsum(x2):
    """fp32 [N] column sums of an [M, N] matrix in the compute dtype (HIP; the final reduction joins the deferred batch)."""
    C.require_gpu(x2)
    M, N = x2.shape
    out = torch.empty(N, dtype=torch.float32, device=x2.device)
    _keep(out)
    ws = _ws(C.lib().tsasr_colsum_workspace_bytes(M, N), x2.device)
    with prof.region("colsum"):
        C.check(C.lib().tsasr_colsum(C.ptr(x2), C.ptr(out), M, N, 0, C.io_dtype(x2), C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_colsum")
    return out


def _out_len(n):
    return (n - 1) // 2 + 1


class _FrontendC1Fn(torch.autograd.Function):
    """Block 1 (one input channel): 3x3 stride-2 conv and 1x1 stride-2 residual conv in one direct HIP kernel."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, causal):
        C.require_gpu(x)
        xc = x.contiguous()
        B, T, Fq = xc.shape
        Co = w1.shape[0]
        f = lambda t: _f32(t).contiguous()  # noqa: E731
        y1 = torch.empty(B, _out_len(T), _out_len(Fq), Co, dtype=xc.dtype, device=xc.device)
        y2 = torch.empty_like(y1)
        with prof.region("frontend_c1_fwd"):
            C.check(C.lib().tsasr_frontend_c1_fwd(C.ptr(xc), C.ptr(f(w1)), C.ptr(f(b1)), C.ptr(f(w2)), C.ptr(f(b2)), C.ptr(y1), C.ptr(y2),
                                                  B, T, Fq, Co, int(causal), C.io_dtype(xc), C.stream_ptr()), "tsasr_frontend_c1_fwd")
        ctx.save_for_backward(xc)
        ctx.cfg = (bool(causal), Co, (w1, b1, w2, b2))
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        (xc,) = ctx.saved_tensors
        causal, Co, prm = ctx.cfg
        B, T, Fq = xc.shape
        dpar = torch.empty(Co * 12, dtype=torch.float32, device=xc.device)
        _keep(dpar)
        ws = _ws(C.lib().tsasr_frontend_c1_bwd_workspace_bytes(Co), xc.device)
        with prof.region("frontend_c1_bwd"):
            C.check(C.lib().tsasr_frontend_c1_bwd(C.ptr(xc), C.ptr(dy1.contiguous()), C.ptr(dy2.contiguous()), C.ptr(dpar), B, T, Fq, Co,
                                                  int(causal), C.io_dtype(xc), C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_frontend_c1_bwd")
        return (None, _pgrad(prm[0], dpar[:Co * 9]), _pgrad(prm[1], dpar[Co * 9:Co * 10]), _pgrad(prm[2], dpar[Co * 10:Co * 11]),
                _pgrad(prm[3], dpar[Co * 11:]), None)


CONV_IMPLICIT = os.environ.get("TSASR_CONV_IMPLICIT", "1") != "0"      # front-end block 2 through the implicit-GEMM kernels (tests / A-B runs switch it off: the im2col path)


CONV_DGRAD_IMPLICIT = os.environ.get("TSASR_CONV_DGRAD", "1") != "0"      # ... and its data gradient as one gathered GEMM (0: dy . Wm + col2im, the A/B path)
_CONV_DGRAD_PLANS = {}


def _conv_dgrad_plan(B, T, Fq, causal, device):
    """Device copy of tsasr_conv3x3s2_dgrad's plan for this shape (host-side function of (B, T, F, causal); built and uploaded on first use - the
    eager warm-up steps - so that a captured step only sees the device pointer). None: a shape the plan does not cover (T or F below 4)."""
    key = (B, T, Fq, bool(causal), str(device))
    if key not in _CONV_DGRAD_PLANS:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("front-end data-gradient plan requested for a new shape during graph capture: run one eager step of this shape first")
        n = C.lib().tsasr_conv3x3s2_dgrad_plan_bytes(B, T, Fq, int(causal))
        if n == 0:      # (an index with more than four contributions: does not occur for T, F >= 2; the HIP GEMM + col2im path takes it)
            _CONV_DGRAD_PLANS[key] = None
        else:
            host = torch.empty(n // 4, dtype=torch.int32)
            C.check(C.lib().tsasr_conv3x3s2_dgrad_plan(B, T, Fq, int(causal), host.data_ptr(), n), "tsasr_conv3x3s2_dgrad_plan")
            _CONV_DGRAD_PLANS[key] = host.to(device)
    return _CONV_DGRAD_PLANS[key]


def _pgrad_view(dwm, Co, Ci, dtype):
    """dWm [Co, (kt, kf, ci)] fp32 -> the reference's [Co, ci, kF, kT] filter gradient."""
    return dwm.view(Co, 3, 3, Ci).permute(0, 3, 2, 1).to(dtype)


def _pgrad_view2(dw2f, shape, dtype):
    return dw2f.view(shape).to(dtype)


class _FrontendConvFn(torch.autograd.Function):
    """A ConvBlock's two convolutions for C_in > 1 (SB/lobes/models/convolution.py:178-266): bf16, C_in in {64, 128}, C_out = 128 = implicit GEMMs
    (csrc/gemm.hip conv_s2_*: the ring kernels' loader waves gather the 3x3 patch rows; forward, filter gradients and - gathering dy rows
    per class of input pixels - the data gradient). Other shapes / fp32: the tap gather (im2col, padding rule folded in) + the HIP GEMMs + col2im."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, causal):
        C.require_gpu(x)
        xc = x.contiguous()
        B, T, Fq, Ci = xc.shape
        Co = w1.shape[0]
        To, Fo = _out_len(T), _out_len(Fq)
        P = B * To * Fo
        # [Co, (kt, kf, ci)] from the reference's [Co, ci, kF, kT]; in bf16 from the arena's bf16 shadow of the parameters (one permuting copy
        # instead of a permuting copy + two casts, at the head of each branch's forward)
        w1s, w2s = (_bf16_weight(w1), _bf16_weight(w2)) if xc.dtype == torch.bfloat16 else (w1, w2)
        wm = w1s.permute(0, 3, 2, 1).reshape(Co, 9 * Ci).to(xc.dtype).contiguous()
        w2m = w2s.reshape(Co, Ci).to(xc.dtype).contiguous()
        centre = 7 if causal else 4                                      # the tap that reads x[2t', 2f']
        hip = xc.dtype == torch.bfloat16 and Ci % 8 == 0 and Co % 8 == 0
        implicit = hip and CONV_IMPLICIT and Co == 128 and Ci in (64, 128)
        if implicit:   # implicit GEMM: the loader waves gather the patch rows - no [P, 9*Ci] matrix in HBM, nothing but x kept for the backward
            y1 = torch.empty(B, To, Fo, Co, dtype=xc.dtype, device=xc.device)
            y2 = torch.empty(B, To, Fo, Co, dtype=xc.dtype, device=xc.device)
            with prof.region("conv3x3s2_fwd", 2.0 * P * Co * 10 * Ci):
                C.check(C.lib().tsasr_conv3x3s2_fwd(C.ptr(xc), C.ptr(wm), C.ptr(_f32(b1).contiguous()), C.ptr(w2m), C.ptr(_f32(b2).contiguous()),
                                                    C.ptr(y1), C.ptr(y2), B, T, Fq, Ci, Co, int(causal), C.stream_ptr()), "tsasr_conv3x3s2_fwd")
            ctx.save_for_backward(xc, wm, w2m)
            ctx.cfg = (bool(causal), (B, T, Fq, Ci), Co, centre, w1.dtype, b1.dtype, w2.dtype, b2.dtype, w2.shape, hip)
            ctx.biases, ctx.weights, ctx.implicit = (b1, b2), (w1, w2), True
            return y1, y2
        A = torch.empty(P, 9 * Ci, dtype=xc.dtype, device=xc.device)
        with prof.region("frontend_im2col"):
            C.check(C.lib().tsasr_frontend_im2col(C.ptr(xc), C.ptr(A), B, T, Fq, Ci, int(causal), C.io_dtype(xc), C.stream_ptr()),
                    "tsasr_frontend_im2col")
        if hip:   # HIP GEMMs: y1 = A . wm^T ; y2 = A[:, centre tap] . w2m^T (strided rows, lda = 9*Ci)
            # bias in the GEMM epilogue (mode 1 with no activation, no dropout): no separate pass over the 41 MB outputs
            y1 = gemm_bf16_fused(A, wm, P, Co, 9 * Ci, 9 * Ci, 9 * Ci, 0, 0, 1, bias=_f32(b1).contiguous()).view(B, To, Fo, Co)
            y2 = gemm_bf16_fused(A[:, centre * Ci:], w2m, P, Co, Ci, 9 * Ci, Ci, 0, 0, 1, bias=_f32(b2).contiguous()).view(B, To, Fo, Co)
        elif xc.dtype == torch.float32 and _F32_HIP_GEMM:   # parity mode: the same two products on the fp32 HIP GEMM
            y1 = (gemm_f32(A, wm, P, Co, 9 * Ci, 9 * Ci, 9 * Ci, 0, 0) + b1.float()).view(B, To, Fo, Co)
            y2 = (gemm_f32(A[:, centre * Ci:], w2m, P, Co, Ci, 9 * Ci, Ci, 0, 0) + b2.float()).view(B, To, Fo, Co)
        else:
            lib_fallback("frontend conv block 2", f"{xc.dtype}, fp32 HIP GEMM off")
            Ac = A.view(P, 9, Ci)[:, centre, :]
            y1 = F.linear(A, wm, b1.to(xc.dtype)).view(B, To, Fo, Co)
            y2 = F.linear(Ac, w2m, b2.to(xc.dtype)).view(B, To, Fo, Co)
        ctx.implicit = False
        ctx.save_for_backward(A, wm, w2m)
        ctx.cfg = (bool(causal), (B, T, Fq, Ci), Co, centre, w1.dtype, b1.dtype, w2.dtype, b2.dtype, w2.shape, hip)
        ctx.biases = (b1, b2)
        return y1, y2

    @staticmethod
    def backward(ctx, dy1, dy2):
        A, wm, w2m = ctx.saved_tensors
        causal, (B, T, Fq, Ci), Co, centre, dw1t, db1t, dw2t, db2t, w2shape, hip = ctx.cfg
        P = B * _out_len(T) * _out_len(Fq)
        g1, g2 = dy1.reshape(P, Co).contiguous(), dy2.reshape(P, Co).contiguous()
        if ctx.implicit:    # A is x itself: the filter gradients gather their patch rows in the loader waves; the data gradient is dA + col2im
            xc = A
            # the 3x3 gradient comes out in the parameter's own layout [Co, ci, kF, kT]: both filter gradients join the arena's batched add
            # (they were a permuting ATen add + a plain one per branch, the speaker branch's at the very end of backward)
            dwm = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device=xc.device)
            dw2f = torch.empty(Co, Ci, dtype=torch.float32, device=xc.device)
            ws = _ws(C.lib().tsasr_conv3x3s2_wgrad_workspace_bytes(B, T, Fq, Ci), xc.device)
            with prof.region("conv3x3s2_wgrad", 2.0 * P * Co * 10 * Ci):
                C.check(C.lib().tsasr_conv3x3s2_wgrad_filters(C.ptr(g1), C.ptr(g2), C.ptr(xc), C.ptr(dwm), C.ptr(dw2f), B, T, Fq, Ci, Co, int(causal),
                                                              C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_conv3x3s2_wgrad_filters")
            db1, db2 = _pgrad(ctx.biases[0], colsum(g1)), _pgrad(ctx.biases[1], colsum(g2))
            dx = torch.empty(B, T, Fq, Ci, dtype=xc.dtype, device=xc.device)
            plan = _conv_dgrad_plan(B, T, Fq, causal, xc.device) if CONV_DGRAD_IMPLICIT else None
            if plan is not None:    # one gathered GEMM per class of input pixels: no [P, 9*Ci] gradient matrix, no inverse gather
                with prof.region("conv3x3s2_dgrad", 2.0 * P * Co * 10 * Ci):
                    C.check(C.lib().tsasr_conv3x3s2_dgrad(C.ptr(g1), C.ptr(g2), C.ptr(wm), C.ptr(w2m), C.ptr(dx), B, T, Fq, Ci, Co, int(causal),
                                                          C.ptr(plan), plan.numel() * 4, C.stream_ptr()), "tsasr_conv3x3s2_dgrad")
            else:
                dA = gemm_bf16(g1, wm, P, 9 * Ci, Co, Co, 9 * Ci, 0, 1)                                         # g1 . wm
                dR = gemm_bf16(g2, w2m, P, Ci, Co, Co, Ci, 0, 1)
                with prof.region("frontend_col2im"):
                    C.check(C.lib().tsasr_frontend_col2im(C.ptr(dA), C.ptr(dR), C.ptr(dx), B, T, Fq, Ci, int(causal), C.io_dtype(xc), C.stream_ptr()),
                            "tsasr_frontend_col2im")
            return dx, _pgrad(ctx.weights[0], dwm.to(dw1t)), db1, _pgrad(ctx.weights[1], dw2f.to(dw2t)), db2, None
        if hip:
            dwm = gemm_bf16(g1, A, Co, 9 * Ci, P, Co, 9 * Ci, 1, 1, out_dtype=torch.float32)              # g1^T . A
            dw2 = gemm_bf16(g2, A[:, centre * Ci:], Co, Ci, P, Co, 9 * Ci, 1, 1, out_dtype=torch.float32).view(w2shape).to(dw2t)
            dA = gemm_bf16(g1, wm, P, 9 * Ci, Co, Co, 9 * Ci, 0, 1)                                         # g1 . wm
            dR = gemm_bf16(g2, w2m, P, Ci, Co, Co, Ci, 0, 1)
        elif A.dtype == torch.float32 and _F32_HIP_GEMM:
            dwm = gemm_f32(g1, A, Co, 9 * Ci, P, Co, 9 * Ci, 1, 1)
            dw2 = gemm_f32(g2, A[:, centre * Ci:], Co, Ci, P, Co, 9 * Ci, 1, 1).view(w2shape).to(dw2t)
            dA = gemm_f32(g1, wm, P, 9 * Ci, Co, Co, 9 * Ci, 0, 1)
            dR = gemm_f32(g2, w2m, P, Ci, Co, Co, Ci, 0, 1)
        else:
            dwm = g1.t() @ A
            dw2 = (g2.t() @ A.view(P, 9, Ci)[:, centre, :]).view(w2shape).to(dw2t)
            dA = g1 @ wm
            dR = g2 @ w2m
        dw1 = dwm.view(Co, 3, 3, Ci).permute(0, 3, 2, 1).to(dw1t)
        if hip:   # column sums on a HIP kernel, added to the arena by the batched end-of-backward add
            db1, db2 = _pgrad(ctx.biases[0], colsum(g1)), _pgrad(ctx.biases[1], colsum(g2))
        else:
            db1, db2 = g1.sum(0, dtype=torch.float32).to(db1t), g2.sum(0, dtype=torch.float32).to(db2t)
        dx = torch.empty(B, T, Fq, Ci, dtype=A.dtype, device=A.device)
        with prof.region("frontend_col2im"):
            C.check(C.lib().tsasr_frontend_col2im(C.ptr(dA), C.ptr(dR), C.ptr(dx), B, T, Fq, Ci, int(causal), C.io_dtype(A), C.stream_ptr()),
                    "tsasr_frontend_col2im")
        return dx, dw1, db1, dw2, db2, None


class _FrontendBlockFn(torch.autograd.Function):
    """A whole ConvBlock in one kernel per direction (csrc/frontend_block.hip): out = Drop(LN_r(r) + Drop(LeakyReLU(LN_y(y)))).
    ``feats`` [B,T,F] given (block 1): (y, r) are convolved on the fly and recomputed in the backward; else (y, r) = ``y1``, ``y2``
    (the im2col GEMM outputs of a wider block, input sizes ``tin``, ``fin``) and the backward returns their gradients."""

    @staticmethod
    def forward(ctx, feats, y1, y2, conv_params, ln_params, tin, fin, causal, slope, eps, p_inner, seed_inner, p_outer, seed_outer,
                *flat_params):
        conv = feats is not None
        src = feats if conv else y1
        C.require_gpu(src)
        f = lambda t: _f32(t).contiguous()  # noqa: E731
        if conv:
            xc = feats.contiguous()
            B, T, Fq = xc.shape
            Co = conv_params[0].shape[0]
            y1c = y2c = None
            cw = [f(t) for t in conv_params]
        else:
            xc = None
            y1c, y2c = y1.contiguous(), y2.contiguous()
            B, T, Fq, Co = y1c.shape[0], int(tin), int(fin), y1c.shape[-1]
            cw = [None] * 4
        To, Fo = _out_len(T), _out_len(Fq)
        ln = [f(t).reshape(-1) for t in ln_params]
        ref = xc if conv else y1c
        out = torch.empty(B, To, Fo, Co, dtype=ref.dtype, device=ref.device)
        stats = torch.empty(B * To, 4, dtype=torch.float32, device=ref.device)
        with prof.region("frontend_block_fwd"):
            C.check(C.lib().tsasr_frontend_block_fwd(C.ptr(xc), C.ptr(y1c), C.ptr(y2c), C.ptr(cw[0]), C.ptr(cw[1]), C.ptr(cw[2]), C.ptr(cw[3]),
                                                     C.ptr(ln[0]), C.ptr(ln[1]), C.ptr(ln[2]), C.ptr(ln[3]), C.ptr(out), C.ptr(stats),
                                                     B, T, Fq, Co, int(causal), float(slope), float(eps), float(p_inner), seed_inner,
                                                     float(p_outer), seed_outer, C.ptr(seed_state(ref.device)), C.io_dtype(ref),
                                                     C.stream_ptr()), "tsasr_frontend_block_fwd")
        ctx.save_for_backward(xc, y1c, y2c, stats, *cw, *ln[:3])
        ctx.cfg = (conv, B, T, Fq, Co, bool(causal), float(slope), float(p_inner), seed_inner, float(p_outer), seed_outer,
                   conv_params, ln_params)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, y1c, y2c, stats, w1, b1, w2, b2, g1, be1, g2 = ctx.saved_tensors
        conv, B, T, Fq, Co, causal, slope, p1, s1, p2, s2, conv_params, ln_params = ctx.cfg
        To, Fo = _out_len(T), _out_len(Fq)
        dout = dout.contiguous()
        n = C.lib().tsasr_frontend_block_dparams(Fo, Co, int(conv))
        dpar = torch.empty(n, dtype=torch.float32, device=dout.device)
        _keep(dpar)
        dy1 = dy2 = None
        if not conv:
            dy1, dy2 = torch.empty_like(y1c), torch.empty_like(y2c)
        ws = _ws(C.lib().tsasr_frontend_block_bwd_workspace_bytes(Fo, Co, int(conv)), dout.device)
        with prof.region("frontend_block_bwd"):
            C.check(C.lib().tsasr_frontend_block_bwd(C.ptr(xc), C.ptr(y1c), C.ptr(y2c), C.ptr(dout), C.ptr(w1), C.ptr(b1), C.ptr(w2), C.ptr(b2),
                                                     C.ptr(g1), C.ptr(be1), C.ptr(g2), C.ptr(stats), C.ptr(dy1), C.ptr(dy2), C.ptr(dpar),
                                                     B, T, Fq, Co, int(causal), slope, p1, s1, p2, s2, C.ptr(seed_state(dout.device)),
                                                     C.io_dtype(dout), C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_frontend_block_bwd")
        o = 0
        grads = []
        if conv:
            for prm, width in zip(conv_params, (Co * 9, Co, Co, Co)):
                grads.append(_pgrad(prm, dpar[o:o + width]))
                o += width
        for prm in ln_params:
            grads.append(_pgrad(prm, dpar[o:o + Fo * Co]))
            o += Fo * Co
        if not conv:
            grads = [None] * 4 + grads
        return (None, dy1, dy2, None, None, None, None, None, None, None, None, None, None, None, *grads)


def frontend_block_supported(x, out_channels):
    """True when the fused ConvBlock kernels take this block (bf16/fp32 on the GPU, 128 output channels, F' <= 40)."""
    return x.is_cuda and bool(C.lib().tsasr_frontend_block_supported(_out_len(x.shape[2]), int(out_channels)))


def frontend_block(x, conv_w, conv_b, red_w, red_b, ln_w, ln_b, rln_w, rln_b, padding, slope, eps, p, training):
    """ConvBlock forward: Dropout_p(LN(conv1x1_s2(x)) + Dropout_p(LeakyReLU(LN(conv3x3_s2(x))))); x [B,T,F,C_in] channels-last."""
    if padding not in ("same", "causal"):
        raise ValueError("Padding must be 'same' or 'causal'. Got " + str(padding))
    p = float(p) if training else 0.0
    s1, s2 = (next_seed(), next_seed()) if p > 0 else (0, 0)
    causal = padding == "causal"
    conv_params, ln_params = (conv_w, conv_b, red_w, red_b), (ln_w, ln_b, rln_w, rln_b)
    if x.shape[-1] == 1:
        return _FrontendBlockFn.apply(x.squeeze(-1), None, None, conv_params, ln_params, 0, 0, causal, slope, eps, p, s1, p, s2,
                                      *conv_params, *ln_params)
    y1, y2 = _FrontendConvFn.apply(x, conv_w, conv_b, red_w, red_b, causal)
    return _FrontendBlockFn.apply(None, y1, y2, None, ln_params, x.shape[1], x.shape[2], causal, slope, eps, p, s1, p, s2,
                                  None, None, None, None, *ln_params)


def frontend_convs(x, w1, b1, w2, b2, padding):
    """x [B,T,F,C] -> (conv3x3_s2(x) + b1, conv1x1_s2(x) + b2), both [B,T',F',C_out]; padding 'same' (reflect) or 'causal'."""
    if padding not in ("same", "causal"):
        raise ValueError("Padding must be 'same' or 'causal'. Got " + str(padding))
    if x.shape[-1] == 1:
        return _FrontendC1Fn.apply(x.squeeze(-1), w1, b1, w2, b2, padding == "causal")
    return _FrontendConvFn.apply(x, w1, b1, w2, b2, padding == "causal")


# ---------------------------------------------------------------------------------------------------------
_ATTN_KEEPBITS = True     # False (tests): the backward hashes the attention dropout mask again instead of reading the forward's keep-bits


class _RelPosAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, pdrop, seed, dpk_deferrable=False):
        C.require_gpu(qkv, pk)
        ctx.dpk_deferrable = bool(dpk_deferrable)
        qkvc, pkc = qkv.contiguous(), pk.contiguous()
        B, T, D3 = qkvc.shape
        D = D3 // 3
        Dh = D // H
        u = _f32(pos_bias_u).reshape(-1).contiguous()   # (Dh,H) storage read as [H,Dh]
        v = _f32(pos_bias_v).reshape(-1).contiguous()
        out = torch.empty(B, T, D, dtype=qkvc.dtype, device=qkvc.device)
        lse = torch.empty(B, H, T, dtype=torch.float32, device=qkvc.device)
        ws_bytes = C.lib().tsasr_relpos_attn_fwd_workspace_bytes(B, T, H)   # > 0: long sequence, small batch - keys split across workgroups
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkvc.device) if ws_bytes else None   # consumed in stream order by the merge launch
        # short sequences: the forward leaves its dropout keep-bits for the backward (32 bytes per query row and head) instead of both hashing
        kb_bytes = (C.lib().tsasr_relpos_attn_keepbits_bytes(B, T, H)
                    if _ATTN_KEEPBITS and pdrop > 0 and Dh == 64 and qkvc.dtype == torch.bfloat16 and ctx.needs_input_grad[0] else 0)
        kb = torch.empty(kb_bytes, dtype=torch.uint8, device=qkvc.device) if kb_bytes else None
        if kb is not None:
            C.lib().tsasr_relpos_attn_keepbits(C.ptr(kb))
        with prof.region("relpos_attn_fwd"):
            C.check(C.lib().tsasr_relpos_attn_fwd_ws(C.ptr(qkvc), C.ptr(pkc), C.ptr(u), C.ptr(v), C.ptr(key_lens), C.ptr(out), C.ptr(lse),
                                                     B, T, H, Dh, float(scale), int(causal), float(pdrop), seed,
                                                     C.ptr(seed_state(qkvc.device)), C.io_dtype(qkvc),
                                                     C.ptr(ws), ws_bytes, C.stream_ptr()), "tsasr_relpos_attn_fwd")
        ctx.save_for_backward(qkvc, pkc, pos_bias_u, pos_bias_v, key_lens, out, lse)
        ctx.keepbits = kb
        ctx.cfg = (H, float(scale), int(causal), float(pdrop), seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkvc, pkc, pu, pv, key_lens, out, lse = ctx.saved_tensors
        H, scale, causal, pdrop, seed = ctx.cfg
        B, T, D3 = qkvc.shape
        D = D3 // 3
        Dh, R = D // H, 2 * T - 1
        u = _f32(pu).reshape(-1).contiguous()
        v = _f32(pv).reshape(-1).contiguous()
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkvc)
        dpk = torch.empty(R, D, dtype=qkvc.dtype, device=qkvc.device)
        du, dv = torch.empty_like(u), torch.empty_like(v)
        _keep(du, dv)
        ws = _ws(C.lib().tsasr_relpos_attn_bwd_workspace_bytes(B, T, H), qkvc.device)
        # d(pk) feeds only linear_pos's weight gradient. When that consumer is the arena's queued grouped launch (the caller says so:
        # pk came out of _LinearFn on a bf16 leaf weight) and reductions are being deferred (the workspace then outlives the step's
        # backward), the pass is queued and runs grouped with every other layer's right before that launch (dpk_flush in wgrad_flush)
        sink = _GRAD_SINK
        defer = (ctx.dpk_deferrable and _DPK["on"] and _DEFER["on"] and sink is not None and getattr(sink, "in_backward", False)
                 and getattr(sink, "collect_wgrads", False) and _WG["ring"] is not None and C.lib().tsasr_relpos_dpk_pending() < _DPK_MAX_JOBS)
        if defer:
            _keep(key_lens)               # the queued pass reads it at the flush, after autograd has released this node's saved tensors
            _DPK["outs"].add(dpk.data_ptr())    # whoever reads dpk outside the grouped weight-gradient launch flushes first (_LinearFn.backward)
        if _DPK["on"] and not defer:      # this call launches its own pass (queued ones first: the switch refuses to go off over a queue)
            dpk_flush_joined()
            C.check(C.lib().tsasr_relpos_dpk_defer(0), "tsasr_relpos_dpk_defer")
        if ctx.keepbits is not None:
            C.lib().tsasr_relpos_attn_keepbits(C.ptr(ctx.keepbits))
        with prof.region("relpos_attn_bwd"):
            C.check(C.lib().tsasr_relpos_attn_bwd(C.ptr(qkvc), C.ptr(pkc), C.ptr(u), C.ptr(v), C.ptr(key_lens), C.ptr(out), C.ptr(dout),
                                                  C.ptr(lse), C.ptr(dqkv), C.ptr(dpk), C.ptr(du), C.ptr(dv), B, T, H, Dh, scale, int(causal),
                                                  pdrop, seed, C.ptr(seed_state(qkvc.device)), C.io_dtype(qkvc), C.ptr(ws), ws.numel(), C.stream_ptr()),
                    "tsasr_relpos_attn_bwd")
        if _DPK["on"] and not defer:
            C.check(C.lib().tsasr_relpos_dpk_defer(1), "tsasr_relpos_dpk_defer")     # (this call launched its pass itself)
        return dqkv, dpk, _pgrad(pu, du), _pgrad(pv, dv), None, None, None, None, None, None, None


ATTN_F32_EXACT = True     # fp32 activations: attention in exact fp32 arithmetic (csrc/attention_f32.hip). False: the MFMA kernels with fp32
                          # storage and bf16-rounded operands (tests that cover that instantiation switch it off)


def _strides9(q, k, v):
    """{batch, row, head} element strides of three [B, T, H, Dh] views with unit stride along Dh (HOST array for tsasr_attn_f32_*)."""
    import ctypes
    out = []
    for t in (q, k, v):
        if t.dim() != 4 or t.stride(3) != 1:
            raise ValueError("attention operands must be [B, T, H, Dh] views with unit stride along Dh")
        out += [t.stride(0), t.stride(1), t.stride(2)]
    return (ctypes.c_longlong * 9)(*out)


class _AttnF32Fn(torch.autograd.Function):
    """softmax(scale * ((q+u).k^T [+ (q+v).p_rel^T]) + masks) . v in exact fp32 arithmetic (tsasr_attn_f32_*), io dtype fp32 or bf16.
    q [B,Tq,H,Dh], k / v [B,Tk,H,Dh] are (possibly strided) views; with ``pk`` [2T-1, H*Dh] it is RelPosMHAXL's core
    (SB/nnet/attention.py:586-633), without it torch.nn.MultiheadAttention's (the `cross_attention` injection). Returns [B,Tq,H*Dh]."""

    @staticmethod
    def forward(ctx, q, k, v, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, pdrop, seed):
        C.require_gpu(q, k, v)
        B, Tq, _, Dh = q.shape
        Tk, D = k.shape[1], H * Dh
        st = _strides9(q, k, v)
        u = None if pos_bias_u is None else _f32(pos_bias_u).reshape(-1).contiguous()     # (Dh,H) storage read as [H,Dh]
        vb = None if pos_bias_v is None else _f32(pos_bias_v).reshape(-1).contiguous()
        pkc = None if pk is None else pk.contiguous()
        out = torch.empty(B, Tq, D, dtype=q.dtype, device=q.device)
        lse = torch.empty(B, H, Tq, dtype=torch.float32, device=q.device)
        with prof.region("attn_f32_fwd"):
            C.check(C.lib().tsasr_attn_f32_fwd(C.ptr(q), C.ptr(k), C.ptr(v), st, C.ptr(pkc), C.ptr(u), C.ptr(vb), C.ptr(key_lens), C.ptr(out),
                                               C.ptr(lse), B, Tq, Tk, H, Dh, float(scale), int(causal), float(pdrop), seed,
                                               C.ptr(seed_state(q.device)), C.io_dtype(q), C.stream_ptr()), "tsasr_attn_f32_fwd")
        ctx.save_for_backward(q, k, v, pkc, pos_bias_u, pos_bias_v, key_lens, out, lse)
        ctx.cfg = (H, float(scale), int(causal), float(pdrop), seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, pkc, pu, pv, key_lens, out, lse = ctx.saved_tensors
        H, scale, causal, pdrop, seed = ctx.cfg
        B, Tq, _, Dh = q.shape
        Tk = k.shape[1]
        st = _strides9(q, k, v)
        dq, dk, dv = (torch.empty(t.shape, dtype=t.dtype, device=t.device) for t in (q, k, v))     # contiguous [B,T,H,Dh]
        dst = _strides9(dq, dk, dv)
        dout = dout.contiguous()
        u = None if pu is None else _f32(pu).reshape(-1).contiguous()
        vb = None if pv is None else _f32(pv).reshape(-1).contiguous()
        dpk = None if pkc is None else torch.empty_like(pkc)
        du = None if pu is None else torch.empty(H * Dh, dtype=torch.float32, device=q.device)
        dvb = None if pv is None else torch.empty(H * Dh, dtype=torch.float32, device=q.device)
        _keep(du, dvb)
        ws = _ws(C.lib().tsasr_attn_f32_bwd_workspace_bytes(B, Tq, Tk, H, Dh), q.device)
        with prof.region("attn_f32_bwd"):
            C.check(C.lib().tsasr_attn_f32_bwd(C.ptr(q), C.ptr(k), C.ptr(v), st, C.ptr(pkc), C.ptr(u), C.ptr(vb), C.ptr(key_lens), C.ptr(out),
                                               C.ptr(dout), C.ptr(lse), C.ptr(dq), C.ptr(dk), C.ptr(dv), dst, C.ptr(dpk), C.ptr(du), C.ptr(dvb),
                                               B, Tq, Tk, H, Dh, scale, causal, pdrop, seed, C.ptr(seed_state(q.device)), C.io_dtype(q),
                                               C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_attn_f32_bwd")
        return (dq, dk, dv, dpk, None if pu is None else _pgrad(pu, du), None if pv is None else _pgrad(pv, dvb), None, None, None, None,
                None, None)


def attention_f32(q, k, v, H, scale, key_lens=None, causal=False, dropout_p=0.0, pk=None, pos_bias_u=None, pos_bias_v=None):
    """Exact-fp32 attention on [B,T,H,Dh] views (see _AttnF32Fn); returns the context as [B,Tq,H*Dh]."""
    p = float(dropout_p)
    return _AttnF32Fn.apply(q, k, v, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, p, next_seed() if p > 0 else 0)


def _relpos_attention_f32(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, dropout_p):
    """RelPosMHAXL's core on the interleaved qkv [B,T,H*(Q|K|V)*Dh] tensor through the exact-fp32 kernels."""
    B, T, D3 = qkv.shape
    Dh = D3 // 3 // H
    x = qkv.view(B, T, H, 3 * Dh) if qkv.is_contiguous() else qkv.contiguous().view(B, T, H, 3 * Dh)
    return attention_f32(x[..., :Dh], x[..., Dh:2 * Dh], x[..., 2 * Dh:], H, scale, key_lens, causal, dropout_p, pk, pos_bias_u, pos_bias_v)


def relpos_attention(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, dropout_p, need_weights, dpk_deferrable=False):
    """Fused HIP kernels (forward and backward) unless the caller wants the [B,H,T,T] weights back (plots only). ``dpk_deferrable``:
    the caller made ``pk`` with ops.matmul_nt on the HIP GEMM path and nothing else reads its gradient (see _RelPosAttnFn.backward)."""
    if need_weights:
        lib_fallback("relpos_attention(need_weights)", "the [B,H,T,T] attention map is only materialised by the ATen route")
        return _relpos_attention_glue(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, dropout_p, need_weights)
    p = float(dropout_p)
    if ATTN_F32_EXACT and qkv.dtype == torch.float32:      # compute_dtype fp32 = the parity mode: no operand is rounded to bf16
        return _relpos_attention_f32(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, p), None
    return _RelPosAttnFn.apply(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, p, next_seed() if p > 0 else 0, dpk_deferrable), None


def _relpos_attention_glue(qkv, pk, pos_bias_u, pos_bias_v, key_lens, H, scale, causal, dropout_p, need_weights):
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
    if causal:   # 1 / True: look-ahead mask; C > 1: block-causal chunks of C frames (build extension, csrc/attention.hip)
        lim = idx if int(causal) <= 1 else (idx // int(causal) + 1) * int(causal) - 1
        score = score.masked_fill(idx[None, :] > lim[:, None], float("-inf"))
    if key_lens is not None:
        score = score.masked_fill((idx[None, :] >= key_lens[:, None]).view(B, 1, 1, T), float("-inf"))
    attn = torch.softmax(score, dim=-1)
    pa = F.dropout(attn, dropout_p, True) if dropout_p > 0 else attn
    o = torch.matmul(pa.to(v.dtype), v.transpose(1, 2)).transpose(1, 2).reshape(B, T, D)
    return o, (attn if need_weights else None)


class _ConvModCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y2, b2, conv_w, conv_b, ln_w, ln_b, causal, eps, slope):
        C.require_gpu(y2)
        y2c = y2.contiguous()
        B, T, D2 = y2c.shape
        D, K = D2 // 2, conv_w.shape[-1]
        f = lambda t: None if t is None else _f32(t).contiguous()  # noqa: E731
        b2f, cw, cb, g, be = f(b2), f(conv_w).reshape(D, K), f(conv_b), f(ln_w), f(ln_b)
        z = torch.empty(B, T, D, dtype=y2c.dtype, device=y2c.device)
        c_save = torch.empty_like(z)
        mean = torch.empty(B * T, dtype=torch.float32, device=y2c.device)
        rstd = torch.empty_like(mean)
        with prof.region("convmod_fwd"):
            C.check(C.lib().tsasr_convmod_fwd(C.ptr(y2c), C.ptr(b2f), C.ptr(cw), C.ptr(cb), C.ptr(g), C.ptr(be), C.ptr(z), C.ptr(c_save),
                                              C.ptr(mean), C.ptr(rstd), B, T, D, K, int(bool(causal)), float(eps), float(slope),
                                              C.io_dtype(y2c), C.stream_ptr()), "tsasr_convmod_fwd")
        ctx.save_for_backward(y2c, b2f, cw, g, be, c_save, mean, rstd)
        ctx.cfg = (bool(causal), float(slope), b2 is not None, conv_w.shape, (conv_w, conv_b, ln_w, ln_b), b2)
        return z

    @staticmethod
    def backward(ctx, dz):
        y2c, b2f, cw, g, be, c_save, mean, rstd = ctx.saved_tensors
        causal, slope, has_b2, wshape, prm, b2p = ctx.cfg
        B, T, D2 = y2c.shape
        D, K = D2 // 2, cw.shape[-1]
        dz = dz.contiguous()
        dy2 = torch.empty_like(y2c)
        dpar = torch.empty(D * (K + 5), dtype=torch.float32, device=y2c.device)
        _keep(dpar)
        ws = _ws(C.lib().tsasr_convmod_bwd_workspace_bytes(B, T, D, K), y2c.device)
        with prof.region("convmod_bwd"):
            C.check(C.lib().tsasr_convmod_bwd(C.ptr(dz), C.ptr(y2c), C.ptr(b2f), C.ptr(cw), C.ptr(g), C.ptr(be), C.ptr(c_save), C.ptr(mean),
                                              C.ptr(rstd), C.ptr(dy2), C.ptr(dpar), B, T, D, K, int(causal), slope, C.io_dtype(y2c),
                                              C.ptr(ws), ws.numel(), C.stream_ptr()), "tsasr_convmod_bwd")
        dg, dbe, dcb, db2, dcw = dpar[:D], dpar[D:2 * D], dpar[2 * D:3 * D], dpar[3 * D:5 * D], dpar[5 * D:]
        return (dy2, _pgrad(b2p, db2) if has_b2 else None, _pgrad(prm[0], dcw), _pgrad(prm[1], dcb), _pgrad(prm[2], dg),
                _pgrad(prm[3], dbe), None, None, None)


def convmod_core(y2, b2, conv_w, conv_b, ln_w, ln_b, causal, eps, slope):
    """y2 [B,T,2D] (bottleneck GEMM output, bias b2 folded in here) -> LeakyReLU(LN(depthwise_conv(GLU(y2 + b2)))) [B,T,D]."""
    return _ConvModCoreFn.apply(y2, b2, conv_w, conv_b, ln_w, ln_b, causal, eps, slope)


def glu_dwconv_ln_act(y2, conv_w, conv_b, ln_w, ln_b, causal, eps, slope):
    return convmod_core(y2, None, conv_w, conv_b, ln_w, ln_b, causal, eps, slope)
