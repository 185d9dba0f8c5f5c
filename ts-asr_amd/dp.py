"""Data parallelism for the TS-ASR training step: one process per GPU, RCCL over xGMI through torch.distributed.

Reference: speechbrain/core.py:1464-1484 wraps EACH trainable module in its own DistributedDataParallel (9 reducers
for the scratch recipe, several with sub-MB payloads), speechbrain/utils/distributed.py:123-201 initialises the
process group. Here instead:

  * ``GradArena`` owns ONE flat fp32 buffer for the gradients (and one for the parameters) of every module;
    ``p.grad`` / ``p.data`` are views into it. Parameters are laid out in the order backward finishes them
    (recorded on the first step, like DDP's bucket rebuild), cut into a few large buckets (default 32 MiB: xGMI is
    point-to-point, 7 links x ~153 GB/s per GPU, so a collective is per-link bound and wants few large messages).
  * a bucket's all-reduce (average) is launched on RCCL's stream the moment its last gradient has been accumulated
    (post-accumulate-grad hooks) and overlaps with the rest of backward; ``finish_backward`` only makes the compute
    stream wait for the communication stream.
  * grad accumulation: with ``sync_enabled = False`` (Brain.no_sync) nothing is sent, exactly like the reference's
    ``require_backward_grad_sync = False`` (core.py:1585-1615).
No data-path collective other than this one exists on the path.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def ddp_init_group(run_opts):
    """speechbrain/utils/distributed.py:123-201: env:// rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    if not run_opts.get("distributed_launch", False) or is_initialized():
        return
    if "RANK" not in os.environ or "LOCAL_RANK" not in os.environ:
        raise ValueError("To use DDP backend, start your script with:\n\tpython -m torch.distributed.run "
                         "--nproc-per-node=N train.py hparams.yaml --distributed_launch --distributed_backend=nccl")
    backend = run_opts.get("distributed_backend", "nccl")
    if backend not in ("nccl", "gloo", "mpi"):
        raise ValueError(backend + " communication protocol doesn't exist.")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dist.init_process_group(backend=backend)


_DIRECT = {"ranks": 0}


def direct_rccl_init(world, rank, device, force=False):
    """One RCCL communicator per process through the C-ABI (csrc/comm.hip), next to torch.distributed's process group: the unique id
    is made on rank 0 and travels through the process group (one 128-byte broadcast). Returns the communicator's rank count, 0 when
    the backend is not RCCL ("gloo" rehearsals keep torch.distributed collectives) or TSASR_RCCL_DIRECT=0."""
    if _DIRECT["ranks"]:
        return _DIRECT["ranks"]
    if os.environ.get("TSASR_RCCL_DIRECT", "1") == "0" or torch.device(device).type != "cuda":
        return 0
    if not force and not (is_initialized() and dist.get_backend() == "nccl"):
        return 0
    from . import _capi as C
    lib = C.lib()
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")   # the copy PyTorch ships (already mapped once RCCL is in use)
    C.check(lib.tsasr_allreduce_load(path.encode() if os.path.exists(path) else None), "tsasr_allreduce_load")
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        C.check(lib.tsasr_allreduce_unique_id(C.ptr(uid)), "tsasr_allreduce_unique_id")
    if world > 1:
        box = [uid.numpy().tobytes()]
        dist.broadcast_object_list(box, src=0)
        uid = torch.frombuffer(bytearray(box[0]), dtype=torch.uint8).clone()
    torch.cuda.set_device(torch.device(device))
    C.check(lib.tsasr_allreduce_init(C.ptr(uid), world, rank), "tsasr_allreduce_init")
    _DIRECT["ranks"] = world
    return world


def direct_capture_probe(device, world, rank, group=None):
    """True when an all-reduce of the direct communicator, captured into a hipGraph on a forked stream and replayed, averages correctly
    on EVERY rank. Called once before the first multi-rank capture of a step: a capture that fails half-way through a training step
    would leave the arena's queues in an unknown state, a failed probe leaves nothing behind - the step is then captured without
    collectives (one all-reduce between replay and optimizer: round 1's form). The verdict is agreed over the process group (MIN)."""
    import sys
    from . import _capi as C

    def agree(flag_value):
        """MIN over the ranks: every rank takes the same branch next (a rank that replays a collective alone would wait for ever)."""
        if is_initialized() and world > 1:
            flag = torch.tensor([int(flag_value)], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item())
        return int(flag_value)

    ok, g, buf = 1, None, None
    try:
        buf = torch.full((4096,), float(rank + 1), dtype=torch.float32, device=device)
        comm = torch.cuda.Stream(device=device)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            comm.wait_stream(cur)
            C.check(C.lib().tsasr_allreduce_bucket(C.ptr(buf), buf.numel(), C.F32, 1, ctypes.c_void_p(comm.cuda_stream)), "tsasr_allreduce_bucket")
            cur.wait_stream(comm)
    except Exception as e:   # noqa: BLE001 - any failure means "do not capture collectives"
        print(f"[ts-asr_amd] rank {rank}: capturing an RCCL all-reduce failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        ok = 0
    # first agreement: did EVERY rank capture? Only then is the collective replayed (a replay on some ranks only never returns)
    if not agree(ok):
        return False
    try:
        g.replay()
        torch.cuda.synchronize()
        n = _DIRECT["ranks"] or world    # ranks of the direct communicator (the one-rank GPU test tells the arena there are two)
        ok = int(bool(torch.allclose(buf, torch.full_like(buf, (n + 1) / 2.0), rtol=1e-6)))
        if not ok:
            print(f"[ts-asr_amd] rank {rank}: captured all-reduce replayed a wrong average ({float(buf[0])})", file=sys.stderr, flush=True)
    except Exception as e:   # noqa: BLE001
        print(f"[ts-asr_amd] rank {rank}: replaying a captured RCCL all-reduce failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        ok = 0
    return bool(agree(ok))      # second agreement: the replayed value was right everywhere


_PAD = 64     # elements: 128-byte lines of the bf16 shadow, 256 bytes of the fp32 arena (ms per step at 8 / 64 / 128: 11.47 / 11.39 / 11.40)


def _pad8(n):
    return (int(n) + _PAD - 1) // _PAD * _PAD


class GradArena:
    def __init__(self, modules, world_size=1, bucket_bytes=None, group=None):
        if bucket_bytes is None:    # 32 MiB buckets (6-7 per step at 51 M parameters); TSASR_BUCKET_MB for tests with small models
            bucket_bytes = int(float(os.environ.get("TSASR_BUCKET_MB", "32")) * (1 << 20))
        seen, params = set(), []
        for p in modules.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("no trainable parameters")
        self.params_ordered = params
        self.params_initial = list(params)          # construction order: the same on every rank (index space of _agree_order)
        self.world_size, self.group, self.bucket_bytes = world_size, group, bucket_bytes
        # payload of the gradient all-reduce: "fp32" (the reference's DDP) or "bf16" (half the xGMI bytes: a bucket is rounded to bf16,
        # averaged, and written back; the arena and the optimizer stay fp32)
        self.comm_dtype = os.environ.get("TSASR_ALLREDUCE_DTYPE", "fp32")
        # direct RCCL communicator (csrc/comm.hip) when the process group's backend is RCCL: the bucket all-reduces are then plain launches
        # on self.comm_stream - the form that can be captured into the step's hipGraph; else torch.distributed (gloo rehearsals, CPU tests)
        self.direct = bool(world_size > 1 and direct_rccl_init(world_size, dist.get_rank() if is_initialized() else 0, params[0].device))
        self.comm_stream = None
        self.device = params[0].device
        # every parameter starts at a multiple of _PAD (64) elements: its bf16 shadow (and transposed copy) is then line-aligned - what the
        # GEMMs' 16-byte operand pieces and the vectorised transpose want (a 29-element bias early in the arena used to shift every
        # weight behind it off that alignment). The pad words stay zero (parameters and gradients alike).
        self.numel = sum(_pad8(p.numel()) for p in params)
        self.grads = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.flat_params = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        # bf16 shadow of every parameter (MFMA GEMM operand copy); the fused optimizer rewrites it in the same pass
        self.flat_params16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device) if self.device.type == "cuda" else None
        self.in_backward = False
        self.sync_enabled = True
        self._sync_this_step = False
        self._order_seen, self._order_final = [], False
        # gradient contributions per parameter and backward pass (a weight used by two GEMMs - the two halves of the `cat` injection's
        # projection - reports twice): recorded on the unordered first pass; a bucket is complete when ALL of them are in. Counting
        # parameters instead sent layer 0's bucket one contribution early and the late one landed on top of the averaged gradient -
        # ranks drifted apart (tests/helpers/dp_gloo_gpu_check.py)
        self._contrib, self._contrib_step = {}, {}
        self._handles, self.sent_log, self._next_send = [], [], 0
        self.probe = None       # tests: a dict that _send / finish_backward fill with the overlap evidence of one eager step
        self.aux_streams, self._main_stream = [], None
        self.companions = []  # flat buffers that must follow a re-layout (optimizer moments)
        self._deferred, self._keepalive, self._defer_ring = [], None, None
        self.collect_wgrads = self.device.type == "cuda"   # weight-gradient GEMMs are queued and run as one grouped launch per flush
        self._wgrad_hold, self._wgrad_held = [], False
        self._reorder_pending = False
        self._layout(params, first=True)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        if self.device.type == "cuda":
            # pinned / device job tables are created NOW: their first use may be inside a hipGraph capture (multi-rank runs only
            # defer inside the captured step), where a host allocation is not permitted
            from . import ops
            self._defer_ring = ops.TableRing(max(1024, 2 * len(params)) * 20, self.device)
            ops.reduce_defer_prepare(self.device)

    # ---- layout ---------------------------------------------------------------------------------------
    def _layout(self, params, first=False):
        """Place parameters (and their gradients) contiguously in the given order; cut buckets. On a re-layout the
        parameter values and every registered companion buffer (optimizer moments) are permuted alike."""
        new_off, off = {}, 0
        for p in params:
            new_off[id(p)] = off
            off += _pad8(p.numel())
        if first:
            for p in params:
                o, n = new_off[id(p)], p.numel()
                self.flat_params[o:o + n].copy_(p.data.reshape(-1).float())
        else:
            for buf in [self.flat_params] + self.companions:
                old = buf.clone()
                buf.zero_()              # (the pad words between parameters stay zero)
                for p in params:
                    o, n, oo = new_off[id(p)], p.numel(), self.offset[id(p)]
                    buf[o:o + n].copy_(old[oo:oo + n])
        self.offset = new_off
        for p in params:
            o, n = new_off[id(p)], p.numel()
            p.data = self.flat_params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
        self.params_ordered = list(params)
        self.refresh_shadow()
        self.buckets, start, pend = [], 0, []
        limit = max(1, self.bucket_bytes // 4)
        for p in params:
            pend.append(id(p))
            end = self.offset[id(p)] + p.numel()
            if end - start >= limit:
                self.buckets.append({"lo": start, "hi": end, "ids": set(pend), "left": len(pend)})
                start, pend = end, []
        if pend:
            self.buckets.append({"lo": start, "hi": self.numel, "ids": set(pend), "left": len(pend)})
        self.bucket_of = {i: b for b in self.buckets for i in b["ids"]}

    def refresh_shadow(self):
        """(Re)build the bf16 shadow and hand every parameter its view (``p._bf16``; valid while ``p._version`` is unchanged).
        Matrices also get a TRANSPOSED bf16 copy (``p._bf16_t`` [cols, rows]): the k-contiguous operand of dx = dy . W."""
        if self.flat_params16 is None:
            return
        self.flat_params16.copy_(self.flat_params)
        jobs, t_off, tiles = [], 0, 0
        mats = [p for p in self.params_ordered if (p.dim() == 2 or (p.dim() == 3 and p.shape[2] == 1)) and min(p.shape[:2]) >= 16 and p.numel() >= 4096]
        total_t = sum(p.numel() for p in mats)
        if getattr(self, "flat_params16_t", None) is None or self.flat_params16_t.numel() != total_t:
            self.flat_params16_t = torch.empty(max(total_t, 1), dtype=torch.bfloat16, device=self.device)
        for p in self.params_ordered:
            o, n = self.offset[id(p)], p.numel()
            p._bf16 = self.flat_params16[o:o + n].view(p.shape)
            p._bf16_ver = p._version
            p._bf16_t = None
        for p in mats:
            r, c = p.shape[0], p.shape[1]
            jobs.append((self.offset[id(p)], t_off, r, c, tiles))
            p._bf16_t = self.flat_params16_t[t_off:t_off + r * c].view(c, r)
            t_off += r * c
            tiles += ((r + 63) // 64) * ((c + 63) // 64)
        self._tr_njobs, self._tr_tiles = len(jobs), tiles
        self._tr_jobs = torch.tensor(jobs, dtype=torch.int32).reshape(-1).to(self.device) if jobs else None
        self.refresh_transposed()

    def sync_shadow(self):
        """After an optimizer that updates the fp32 parameters through torch (anything but the fused AdamW kernel, which writes the
        shadow itself): rewrite the bf16 shadow and the transposed copies - two launches, graph-capturable. Without it an eager step
        notices the stale shadow through the parameters' version counters and casts on the fly, but a captured step has that
        decision frozen at capture time and kept multiplying with the weights of the capture step (found with SGD under hipGraph)."""
        if self.flat_params16 is None:
            return
        self.flat_params16.copy_(self.flat_params)
        for p in self.params_ordered:
            p._bf16_ver = p._version
        self.refresh_transposed()

    def refresh_transposed(self):
        """Rewrite the transposed weight copies from the bf16 shadow: one launch, issued after every optimizer step (graph-capturable)."""
        if getattr(self, "_tr_jobs", None) is None:
            return
        from . import _capi as C
        C.check(C.lib().tsasr_transpose_many_bf16(C.ptr(self.flat_params16), C.ptr(self.flat_params16_t), C.ptr(self._tr_jobs),
                                                  self._tr_njobs, self._tr_tiles, C.stream_ptr()), "tsasr_transpose_many_bf16")

    # ---- gradient sink: kernels that add a weight gradient straight into the arena (ops._LinearFn) ---------
    def accepts(self, p):
        return self.in_backward and id(p) in self.offset and p.grad is not None

    def mark_ready(self, p):
        self._on_grad(p)

    def defer_add(self, p, g):
        """Small fp32 parameter gradients are collected and added into the arena by ONE kernel at the end of backward
        (instead of one AccumulateGrad add kernel each). Not used on steps that overlap the all-reduce with backward
        (a bucket may only be sent once all its gradients are in)."""
        if self._sync_this_step or not self.accepts(p) or g.dtype != torch.float32 or not g.is_cuda or g.numel() != p.numel():
            return False
        self._deferred.append((g.contiguous(), p))
        return True

    def _flush_deferred(self):
        if not self._deferred:
            return
        import numpy as np
        from . import _capi as C
        n = len(self._deferred)
        from . import ops
        if self._defer_ring is None:            # CPU-constructed arena moved to the GPU later: allocate on first (eager) use
            self._defer_ring = ops.TableRing(max(1024, 2 * n) * 20, self.device)
        k, host, dev, captured = self._defer_ring.acquire()   # ring of pairs eagerly (a pair is rewritten only after its launch ran), one per captured flush
        if host.numel() < n * 20:
            raise RuntimeError("deferred-gradient table too small")
        tab = np.empty(n * 20, np.uint8)
        tab[:n * 8].view(np.uint64)[:] = [g.data_ptr() for g, _ in self._deferred]
        tab[n * 8:n * 16].view(np.uint64)[:] = [p.grad.data_ptr() for _, p in self._deferred]
        tab[n * 16:].view(np.int32)[:] = [g.numel() for g, _ in self._deferred]
        host[:n * 20].copy_(torch.from_numpy(tab))
        if not captured:                        # captured: filled now, uploaded after the capture (upload_captured_tables) - no memcpy node
            dev[:n * 20].copy_(host[:n * 20], non_blocking=True)
        C.check(C.lib().tsasr_accumulate_many(C.ptr(dev), n, C.stream_ptr()), "tsasr_accumulate_many")
        self._defer_ring.launched(k, n * 20)
        self._keepalive = self._deferred    # the temporaries must outlive the launch
        self._deferred = []

    def upload_captured_tables(self):
        if self._defer_ring is not None:
            self._defer_ring.upload_captured()

    def reserve_captured_tables(self, n):
        if self._defer_ring is not None:
            self._defer_ring.reserve_captured(n)

    def abort_backward(self):
        """Error path (a step capture that raised): forget everything queued for this backward pass."""
        self._deferred, self._wgrad_hold, self._wgrad_held, self._handles = [], [], False, []
        self.in_backward = False

    # ---- grouped weight gradients (csrc/wgrad.hip) -------------------------------------------------------
    def wgrad_queued(self, p):
        """ops._wgrad_into queued p's weight gradient. On a step that overlaps the all-reduce with backward, a bucket whose other
        gradients are all in is completed by flushing the queue (one grouped launch), then sent. Otherwise the queue is flushed
        whenever it holds about half a chip's worth of output tiles, on a stream of its own: the grouped launch then runs BESIDE the
        rest of backward (a long kernel on fewer than 256 CUs next to many short, latency-bound ones) instead of after it."""
        if not self._order_final:
            self._order_seen.append(p)
            self._contrib_step[id(p)] = self._contrib_step.get(id(p), 0) + 1
        from . import ops
        if not self._sync_this_step or not self._order_final:
            return
        b = self.bucket_of[id(p)]
        b["left"] -= 1
        b["queued"] = b.get("queued", 0) + 1
        if b["left"] == 0:
            self.flush_wgrads()

    def flush_wgrads(self, hold=False, release=False):
        """Run every queued weight gradient now on the current stream, ordered after every stream of the step. ``hold``: the launch happens
        while other streams of the step are still running (the recipe's early flush under the speaker branch's backward): the operands -
        some were allocated on those streams - stay referenced until the flush from finish_backward (``release``), after the streams
        have joined; dropped here, the caching allocator hands their memory to the other stream's next kernels while this launch still
        reads it."""
        from . import ops
        if ops.wgrad_pending() == 0:
            self._send_completed()      # (an earlier flush may have finished a bucket's queued gradients before its last plain gradient came in)
            return
        self.join_streams()
        if hold:
            ops.wgrad_flush(hold=self._wgrad_hold)
            self._wgrad_held = True
        else:
            ops.wgrad_flush()
            if self._wgrad_held and release:    # held operands go only once every stream of the step has joined (finish_backward)
                self._wgrad_held, self._wgrad_hold = False, []
        self._send_completed()

    def join_streams(self):
        """Order the current stream behind every stream of the step (main + forked branches): what a launch that consumes work queued from
        several streams (grouped weight gradients, deferred d(pk) passes) needs before it goes out."""
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream()
            for st in [self._main_stream] + list(self.aux_streams):
                if st is not None and st != cur:
                    cur.wait_stream(st)

    def _send_completed(self):
        """Buckets whose every contribution is in and whose queued weight gradients have been launched: all-reduce them now."""
        if self._sync_this_step and self._order_final:
            for b in self.buckets:
                if b.get("queued", 0) and b["left"] == 0:
                    b["queued"] = 0
                    self._ready(b)

    def _ready(self, b):
        """Bucket b is complete. Collectives are issued STRICTLY IN BUCKET-INDEX ORDER (as torch DDP's reducer does): a complete bucket
        waits until every lower-index bucket has been sent. Ranks may complete buckets in different orders - an eager step runs the
        predictor's backward right after the joint, a captured step (predictor forked in front of the encoder) after the whole
        mixture encoder, and with length-bucketed batches one rank runs eagerly while another replays - but every rank must issue
        the same sequence of ncclAllReduce calls, or counts mismatch and the ranks hang or average the wrong gradients."""
        b["ready"] = True
        while self._next_send < len(self.buckets) and self.buckets[self._next_send].get("ready"):
            self._send(self.buckets[self._next_send])
            self._next_send += 1

    # ---- per step --------------------------------------------------------------------------------------
    def begin_backward(self, will_sync):
        self.in_backward = True
        self._main_stream = torch.cuda.current_stream() if self.device.type == "cuda" else None
        self._sync_this_step = bool(will_sync) and self.sync_enabled and self.world_size > 1
        if not self._sync_this_step and self.device.type == "cuda":   # (a bucket may only be sent once its gradients are final)
            from . import ops
            ops.reduce_defer_begin(self.device)
            ops.dpk_defer_begin(self.device)      # d(pk) passes of the attention backward: queued, run grouped in front of the weight gradients
        for b in self.buckets:
            b["left"], b["sent"], b["queued"], b["ready"] = sum(self._contrib.get(i, 1) for i in b["ids"]), False, 0, False
        self._next_send = 0
        if not self._order_final:
            self._contrib_step = {}
        self._handles, self.sent_log = [], []

    def _on_grad(self, p):
        if not self._order_final:
            self._order_seen.append(p)
            self._contrib_step[id(p)] = self._contrib_step.get(id(p), 0) + 1
        if not self._sync_this_step or not self._order_final:
            return
        b = self.bucket_of[id(p)]
        b["left"] -= 1
        if b["left"] == 0:
            if b.get("queued", 0):
                self.flush_wgrads()
            else:
                self._ready(b)

    def _send(self, b):
        if b.get("sent"):
            return
        b["sent"] = True
        chunk = self.grads[b["lo"]:b["hi"]]
        if self.device.type == "cuda":
            # the collective is ordered after the CURRENT stream only; a bucket may hold gradients written on another stream
            # of this step (the recipe runs two branches of the model on a second stream, backward follows the forward's streams)
            cur = torch.cuda.current_stream()
            for st in [self._main_stream] + list(self.aux_streams):
                if st is not None and st != cur:
                    cur.wait_stream(st)
        payload = chunk.to(torch.bfloat16) if self.comm_dtype == "bf16" else chunk
        back = chunk if payload is not chunk else None
        if self.direct:   # bare ncclAllReduce on the communication stream (csrc/comm.hip): capturable, no watchdog, ordered by stream joins only
            from . import _capi as C
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=self.device)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            C.check(C.lib().tsasr_allreduce_bucket(C.ptr(payload), payload.numel(), C.BF16 if payload.dtype == torch.bfloat16 else C.F32, 1,
                                                   ctypes.c_void_p(self.comm_stream.cuda_stream)), "tsasr_allreduce_bucket")
            self._handles.append((None, None, payload, back))
            self.sent_log.append((b["lo"], b["hi"]))
            if self.probe is not None and "first_done" not in self.probe and not torch.cuda.is_current_stream_capturing():
                # overlap probe (tests): when was the FIRST bucket's collective issued, and an event behind it on the communication stream
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(self.comm_stream)
                self.probe.update(first_done=ev, first_in_backward=bool(self.in_backward))
            return
        if dist.get_backend(self.group) == "nccl":
            self._handles.append((dist.all_reduce(payload, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None, payload, back))
        else:  # gloo has no AVG
            self._handles.append((dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=self.group, async_op=True), self.world_size, payload, back))
        self.sent_log.append((b["lo"], b["hi"]))

    def finish_backward(self):
        if self.probe is not None and self.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event(enable_timing=True)       # the end of backward on the main stream (every forked stream has been joined)
            ev.record()
            self.probe.update(backward_done=ev, sent_during_backward=len(self.sent_log))
        if self.device.type == "cuda":
            from . import ops
            self.flush_wgrads(release=True)   # every queued weight gradient, one grouped launch (accumulates into the arena)
            if self._wgrad_held:        # nothing was left to flush: the streams have joined (Brain._device_step), release the held operands
                self._wgrad_held, self._wgrad_hold = False, []
            ops.dpk_defer_end()         # (nothing is left unless no weight-gradient launch followed the last attention backward)
            ops.reduce_defer_end()      # every queued partial-sum reduction, one launch
        self._flush_deferred()
        self.in_backward = False
        if self._sync_this_step:
            for b in self.buckets:  # buckets whose parameters got no gradient this step, or first (unordered) step: index order
                self._send(b)
            self._next_send = len(self.buckets)
            if self.direct and self.comm_stream is not None and self._handles:
                torch.cuda.current_stream().wait_stream(self.comm_stream)   # "wait" = one join of the communication stream
            for h, div, payload, back in self._handles:
                if h is not None:
                    h.wait()
                if div is not None:
                    payload.div_(div)
                if back is not None:          # bf16 payload: the averaged bucket goes back into the fp32 arena
                    back.copy_(payload)
            self._handles = []
        if not self._order_final and self._order_seen:
            self._finalize_order()

    def _finalize_order(self):
        """Re-lay the arena in the order gradients became ready on the first backward (grads must be consumed first:
        called from zero_() at the end of the step)."""
        self._reorder_pending = True

    def zero_(self):
        self.grads.zero_()
        if self._reorder_pending:
            seen, order = set(), []
            for p in self._order_seen:
                if id(p) not in seen:
                    seen.add(id(p))
                    order.append(p)
            for p in self.params_ordered:  # parameters that never produced a gradient go last
                if id(p) not in seen:
                    order.append(p)
            order = self._agree_order(order)
            self._layout(order)
            self._order_final, self._reorder_pending, self._order_seen = True, False, []
            self._contrib = dict(self._contrib_step)

    def _agree_order(self, order):
        """Every rank must cut the SAME buckets: rank 0's recorded backward order is broadcast (as indices into the construction
        order, which is identical on every rank) and adopted by all - the reference's DDP does the same when it rebuilds its buckets."""
        if self.world_size <= 1 or not is_initialized():
            return order
        index = {id(p): i for i, p in enumerate(self.params_initial)}
        dev = self.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        idx = torch.tensor([index[id(p)] for p in order], dtype=torch.int64, device=dev)
        dist.broadcast(idx, 0, group=self.group)
        return [self.params_initial[i] for i in idx.cpu().tolist()]

    def broadcast_parameters(self):
        """Rank 0's parameter values to every rank (the reference's DDP constructor does this: SB/core.py:1464-1484)."""
        if self.world_size <= 1 or not is_initialized():
            return
        if dist.get_backend(self.group) == "nccl" or self.device.type == "cpu":
            dist.broadcast(self.flat_params, 0, group=self.group)
        else:
            t = self.flat_params.cpu()
            dist.broadcast(t, 0, group=self.group)
            self.flat_params.copy_(t)
        self.refresh_shadow()

    def allreduce_all(self):
        """Average the whole arena over the ranks (graph mode: between the captured step and the optimizer)."""
        if self.world_size <= 1:
            return
        # nothing can overlap here (the captured step has finished, the optimizer needs every gradient): ONE collective over the whole
        # contiguous arena (204 MB fp32) instead of one per bucket - a ring all-reduce is bandwidth-bound per link, the per-call
        # latency is paid once
        nccl = dist.get_backend(self.group) == "nccl"
        dist.all_reduce(self.grads, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, group=self.group)
        if not nccl:
            self.grads.div_(self.world_size)

    def grad_norm(self):
        return torch.linalg.vector_norm(self.grads)
