"""Training runtime with the ``speechbrain.Brain`` surface the TS-ASR recipes program against
(vendor/speechbrain/speechbrain/core.py:537-1643): ``Brain(modules, opt_class, hparams, run_opts, checkpointer)``,
``compute_forward`` / ``compute_objectives`` overrides, ``fit_batch`` / ``evaluate_batch`` / ``fit`` / ``evaluate``,
``on_stage_start/end``, ``on_fit_batch_end``, ``no_sync``; ``Stage``; ``parse_arguments``.

What is different underneath (MI355X-first):
  * one process per GPU; gradients of ALL modules live in one flat arena that is all-reduced in a few large
    buckets over RCCL/xGMI while backward is still running (dp.GradArena) - the reference wraps each of its
    9 trainable modules in its own DistributedDataParallel reducer (core.py:1464-1484);
  * no host synchronisation inside the step: the loss, the gradient norm and the non-finite check stay on the
    device (the reference does loss.detach().cpu() and grad_norm.item() every micro-batch, core.py:1086,1096);
    ``fit_batch`` returns a device scalar, and the non-finite counter is read back once per ``fit`` epoch/flush;
  * gradient clipping + AdamW run as one pass over the arena (optim.FusedClipAdamW) when opt_class is the
    reference's torch.optim.AdamW partial; any other optimizer class is used as given.
"""
import argparse
import contextlib
import enum
import logging
import os
import sys

import torch

from . import dp as _dp
from . import optim as _optim
from . import prof

logger = logging.getLogger(__name__)


_GRAPH_COMM = os.environ.get("TSASR_GRAPH_COMM", "1") != "0"     # multi-rank graph mode: bucketed all-reduces captured inside the step's graph

class Stage(enum.Enum):
    TRAIN = enum.auto()
    VALID = enum.auto()
    TEST = enum.auto()


class EpochCounter:
    """speechbrain/utils/epoch_loop.py:17-60."""

    def __init__(self, limit):
        self.current, self.limit = 0, int(limit)

    def __iter__(self):
        return self

    def __next__(self):
        if self.current < self.limit:
            self.current += 1
            return self.current
        raise StopIteration


class NoamScheduler:
    """speechbrain/nnet/schedulers.py:363-455: lr = lr0 * sqrt(W) * min(n^-0.5, n * W^-1.5)."""

    def __init__(self, lr_initial, n_warmup_steps, model_size=None):
        self.lr_initial, self.n_warmup_steps = lr_initial, n_warmup_steps
        self.current_lr, self.losses, self.n_steps = lr_initial, [], 0
        self.normalize = n_warmup_steps ** 0.5
        if model_size is not None:
            self.normalize = model_size ** (-0.5)

    def __call__(self, opt):
        self.n_steps += 1
        current_lr = opt.param_groups[0]["lr"]
        lr = self.lr_initial * self._get_lr_scale()
        for group in opt.param_groups:
            group["lr"] = lr
        self.current_lr = current_lr
        return current_lr, lr

    def _get_lr_scale(self):
        n, w = self.n_steps, self.n_warmup_steps
        return self.normalize * min(n ** (-0.5), n * w ** (-1.5))


RUN_OPT_DEFAULTS = {
    "debug": False, "device": "cuda:0", "distributed_launch": False, "distributed_backend": "nccl",
    "find_unused_parameters": False, "auto_mix_prec": False, "bfloat16_mix_prec": False, "max_grad_norm": 5.0,
    "nonfinite_patience": 3, "noprogressbar": True, "ckpt_interval_minutes": 0, "grad_accumulation_factor": 1,
    "optimizer_step_limit": None, "compute_dtype": None,
    # build options (INTEGRATION.md section "Non-finite steps"): False = the reference (a non-finite step is applied, only counted);
    # steps between host reads of the device-side counters (the reference reads every step: 1)
    "skip_nonfinite_step": False, "nonfinite_flush_every": 100,
}


def parse_arguments(arg_list=None):
    """(hparams_file, run_opts, overrides) like speechbrain.core.parse_arguments (core.py:134-393): known run options
    become ``run_opts``; every other ``--key value`` becomes a YAML override. LOCAL_RANK rewrites ``device``."""
    arg_list = list(sys.argv[1:] if arg_list is None else arg_list)
    p = argparse.ArgumentParser(description="Run a TS-ASR experiment on MI355X")
    p.add_argument("param_file", type=str)
    p.add_argument("--debug", default=False, action="store_true")
    p.add_argument("--device", type=str)
    p.add_argument("--distributed_launch", default=False, action="store_true")
    p.add_argument("--distributed_backend", type=str)
    p.add_argument("--find_unused_parameters", default=False, action="store_true")
    p.add_argument("--auto_mix_prec", default=None, action="store_true")
    p.add_argument("--bfloat16_mix_prec", default=None, action="store_true")
    p.add_argument("--max_grad_norm", type=float)
    p.add_argument("--nonfinite_patience", type=int)
    p.add_argument("--skip_nonfinite_step", type=lambda v: str(v).lower() in ("1", "true", "yes"))
    p.add_argument("--nonfinite_flush_every", type=int)
    p.add_argument("--grad_accumulation_factor", type=int)
    p.add_argument("--optimizer_step_limit", type=int)
    p.add_argument("--local_rank", type=int)
    run, rest = p.parse_known_args(arg_list)
    run_opts = {k: v for k, v in vars(run).items() if v is not None and v is not False}
    param_file = run_opts.pop("param_file")
    overrides = {}
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--"):
            raise ValueError(f"unexpected argument {tok!r}")
        if "=" in tok:
            k, v = tok[2:].split("=", 1)
            i += 1
        else:
            k, v = tok[2:], rest[i + 1] if i + 1 < len(rest) else "True"
            i += 2
        import yaml as _yaml
        overrides[k] = _yaml.safe_load(v)
    local_rank = run_opts.pop("local_rank", None)
    if local_rank is None and "LOCAL_RANK" in os.environ:
        local_rank = int(os.environ["LOCAL_RANK"])
    if local_rank is not None and "cuda" in run_opts.get("device", "cuda"):
        run_opts["device"] = f"cuda:{local_rank}"
    return param_file, run_opts, overrides


class Brain:
    def __init__(self, modules=None, opt_class=None, hparams=None, run_opts=None, checkpointer=None, profiler=None):
        self.opt_class, self.checkpointer, self.profiler = opt_class, checkpointer, profiler
        run_opts = dict(run_opts or {})
        hp = dict(hparams or {})
        for k, default in RUN_OPT_DEFAULTS.items():  # CLI > YAML > default, as core.py:587-606
            setattr(self, k, run_opts[k] if k in run_opts else hp.get(k, default))
        self.device = str(self.device)
        if "cuda" in self.device:
            torch.cuda.set_device(int(self.device.split(":")[1]) if ":" in self.device else 0)
        # components without a mirror on the hot path (loggers, ...) arrive as hparams.Unavailable: not modules
        mods = {k: m for k, m in (modules or {}).items() if isinstance(m, torch.nn.Module)}
        self.skipped_modules = sorted(set(modules or {}) - set(mods))
        self.modules = torch.nn.ModuleDict(mods).to(self.device)
        from types import SimpleNamespace
        self.hparams = SimpleNamespace(**hp)
        if self.compute_dtype is None:
            self.compute_dtype = "bf16" if (self.auto_mix_prec and self.bfloat16_mix_prec) else "fp32"
        self.valid_step = self.step = self.optimizer_step = 0
        self.nonfinite_count, self.skipped_steps = 0, 0
        self.nonfinite_flush_every = int(self.nonfinite_flush_every or 0)   # steps between host reads of the device-side non-finite counters
        self._nonfinite_dev = None
        self.avg_train_loss = 0.0
        self.grad_norm_epoch = []
        self.optimizer = None
        self.arena = None
        self._graph_mode, self._graph, self._graph_warmup, self._eager_steps = False, None, 3, 0
        self._static_batches, self._static_loss, self._graphs, self._graph_pool, self._eager_stepped = {}, {}, {}, None, False
        self._graph_comm_ok = None                # multi-rank: may collectives be captured with the step? (probed once, see _graph_comm)
        self._graph_max_shapes, self._seen_shapes = 24, set()
        self._aux_streams = []
        self.rank = int(os.environ.get("RANK", 0))
        self.distributed = bool(self.distributed_launch) and _dp.is_initialized()

    # ---- hooks the recipe overrides ----------------------------------------------------------
    def compute_forward(self, batch, stage):
        raise NotImplementedError

    def compute_objectives(self, predictions, batch, stage):
        raise NotImplementedError

    def on_stage_start(self, stage, epoch=None):
        pass

    def on_stage_end(self, stage, stage_loss, epoch=None):
        pass

    def on_fit_batch_end(self, batch, outputs, loss, should_step):
        pass

    def on_fit_start(self):
        self._setup_dtype()
        self.init_optimizers()

    # ---- setup ---------------------------------------------------------------------------------
    def _setup_dtype(self):
        from . import nnet
        nnet.set_compute_dtype(torch.bfloat16 if self.compute_dtype in ("bf16", torch.bfloat16) else torch.float32)

    def trainable_parameters(self):
        seen, out = set(), []
        for p in self.modules.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def init_optimizers(self):
        if self.opt_class is None or self.optimizer is not None:
            return
        params = self.trainable_parameters()
        # one flat fp32 gradient arena for every module; parameters ordered so that buckets complete in the order
        # backward produces them (dp.GradArena docstring)
        self.arena = _dp.GradArena(self.modules, world_size=_dp.world_size() if self.distributed else 1)
        self.arena.aux_streams = self._aux_streams     # same list object: streams the recipe forks register themselves there
        if self.distributed:
            self.arena.broadcast_parameters()          # identical initial weights on every rank (reference: DDP's constructor broadcast)
        self.optimizer = _optim.make_optimizer(self.opt_class, params, self.arena, self.max_grad_norm, bool(self.skip_nonfinite_step))
        from . import ops as _ops
        _ops.set_grad_sink(self.arena)

    # ---- the training step (core.py:1032-1096) -----------------------------------------------------
    @contextlib.contextmanager
    def no_sync(self, use=True):
        if use and self.arena is not None:
            old = self.arena.sync_enabled
            self.arena.sync_enabled = False
            try:
                yield
            finally:
                self.arena.sync_enabled = old
        else:
            yield

    def fit_batch(self, batch):
        if self.optimizer is None:
            self.on_fit_start()
        self.valid_step += 1
        should_step = (self.valid_step % self.grad_accumulation_factor) == 0
        if self._graph_mode:
            loss, outputs = self._fit_batch_graph(batch, should_step), None
        else:
            with self.no_sync(not should_step):
                loss, outputs = self._device_step(batch, should_step, comm=True)
            if should_step:
                self.optimizer_step += 1
        self.on_fit_batch_end(batch, outputs, loss, should_step)
        return loss

    def _loss_seed(self, loss):
        key = (loss.dtype, loss.device, int(self.grad_accumulation_factor))
        seeds = self.__dict__.setdefault("_loss_seeds", {})
        if key not in seeds:
            seeds[key] = torch.full((), 1.0 / self.grad_accumulation_factor, dtype=loss.dtype, device=loss.device)
        return seeds[key]

    def _device_step(self, batch, should_step, comm):
        """Everything of one micro-batch that runs on the GPU; no host synchronisation, no host->device copies: capturable."""
        from . import ops as _ops
        _ops.begin_step(self.device)
        prof.stamp_begin(self.device)
        prof.stamp("step starts [main]")
        # Every stream a recipe has EVER forked is forked again here, whether this step will use it or not: the joins after backward (below,
        # GradArena.join_streams) wait on all of them, and inside a capture a wait on a stream that is not part of the capture is an error
        # (a stream forked only by some batch shapes - the predictor's own stream for long targets - was joined by the others unforked).
        if torch.device(self.device).type == "cuda":
            cur0 = torch.cuda.current_stream()
            for s in self._aux_streams:
                s.wait_stream(cur0)
        self.arena.begin_backward(should_step and comm)
        outputs = self.compute_forward(batch, Stage.TRAIN)
        loss = self.compute_objectives(outputs, batch, Stage.TRAIN)
        self.check_gradients(loss)
        # d(loss / factor) = 1 / factor: handed to backward() as a constant kept on the device, instead of a division node in front of the
        # loss (SB/core.py:1066 `(loss / self.grad_accumulation_factor).backward()`: three element-wise launches less per step, same bits)
        loss.backward(gradient=self._loss_seed(loss))
        # Streams the recipe forked in forward also ran their share of backward. autograd joins only the streams its LEAF
        # (AccumulateGrad) nodes ran on - and most parameter gradients here bypass those nodes (GEMMs accumulate straight into
        # the arena, small gradients are queued for one batched add) - so join explicitly before anything reads the gradients.
        prof.stamp("backward done [main]")
        for s in self._aux_streams:
            if prof.STAMPS:
                with torch.cuda.stream(s):
                    prof.stamp("backward done [side]")
            torch.cuda.current_stream().wait_stream(s)
        self.arena.finish_backward()          # waits for the overlapped bucket all-reduces (if any), averages over ranks
        prof.stamp("weight gradients and reductions done [main]")
        if should_step and (comm or not self.distributed):
            if comm and not (torch.device(self.device).type == "cuda" and torch.cuda.is_current_stream_capturing()):
                self.optimizer.prepare()      # host half (step count, lr -> device): never inside a capture (_fit_batch_graph does it before a replay)
            self.optimizer.launch()           # clip (global L2 norm, max_grad_norm) + AdamW in one pass over the arena
            self.arena.zero_()
            prof.stamp("optimizer done [main]")
        return loss.detach(), outputs

    # ---- hipGraph replay of the whole step (HIP streams and graphs instead of a tracing compiler) -----------------
    def enable_hip_graph(self, warmup_steps=3, max_shapes=24):
        """After ``warmup_steps`` eager steps (allocator warm, arena laid out in backward order) the step is captured and replayed:
        ~1000 launches collapse into one graph launch. One graph per distinct set of batch tensor shapes (and per flavour, see
        _fit_batch_graph), up to ``max_shapes`` shapes: with length-bucketed batches padded to the bucket edge
        (dataio.DynamicBatchSampler) a real epoch replays a handful of graphs; all graphs share one memory pool (they never run
        concurrently). A shape beyond the cap runs eagerly. With more than one rank the bucketed gradient all-reduces are CAPTURED with
        the step (RCCL kernels as graph nodes on the communication stream, launched when a bucket's gradients are complete, joined before
        the optimizer): the replayed step issues exactly the collectives an eager step issues, in the same order, so ranks may mix eager
        and replayed steps freely (length-bucketed batches: one rank meets a new shape while another replays). TSASR_GRAPH_COMM=0 restores
        round 1's form (ONE un-overlapped all-reduce between the captured step and the optimizer) for A/B runs."""
        self._graph_mode, self._graph_warmup, self._graph_max_shapes = True, int(warmup_steps), int(max_shapes)

    @staticmethod
    def _shape_key(batch):
        key = []
        for k in batch._keys:
            v = getattr(batch, k)
            if isinstance(v, tuple):
                key.append((k,) + tuple(tuple(t.shape) for t in v))
        return tuple(key)

    def _host_draws(self):
        """Random choices the reference makes on the HOST inside compute_forward (SpeedPerturb's speed: speech_augmentation.py:480-493)
        are made here, before the step, and become part of the graph key: one captured graph per (batch shape, speed) - the resampled
        length differs per speed anyway - instead of the capture-time choice replayed for ever."""
        hp = self.hparams
        if (getattr(hp, "augment", False) and not getattr(hp, "input_is_feats", False) and "speed_perturb" in self.modules
                and self.modules.training):
            return (("speed_perturb",) + tuple(self.modules["speed_perturb"].draw()),)
        return ()

    def _fit_batch_graph(self, batch, should_step=True):
        """Two flavours per batch shape: the micro-step that only accumulates gradients and the one that also clips / steps / clears
        (gradient accumulation: grad_accumulation_factor - 1 replays of the first, one of the second)."""
        key = self._shape_key(batch) + self._host_draws()
        full = key not in self._static_batches and len(self._static_batches) >= self._graph_max_shapes
        first = key not in self._seen_shapes      # a new shape runs eagerly once: per-shape caches (positional tables, ...) must be
        self._seen_shapes.add(key)                # filled outside a capture, where their memory would belong to the graph pool
        if self._eager_steps < self._graph_warmup or not self._eager_stepped or full or first:
            # eager until the allocator is warm AND one optimizer step has run: the arena re-lays itself out in backward order when
            # the gradients are first cleared, which must not happen inside a capture
            self._eager_steps += 1
            # Ranks decide eager-vs-replay from their OWN shape history, so both paths must issue the same collectives: with the direct
            # RCCL communicator both send the bucket all-reduces during backward; otherwise (gloo, TSASR_GRAPH_COMM=0) the replay is
            # followed by ONE all-reduce of the whole arena - and so is this eager step
            same_as_replay = self.distributed and not self._graph_comm()
            with self.no_sync(not should_step):
                loss, _ = self._device_step(batch, should_step, comm=not same_as_replay)
            if should_step:
                if same_as_replay:
                    self.arena.allreduce_all()
                    self.optimizer.prepare()
                    self.optimizer.launch()
                    self.arena.zero_()
                self.optimizer_step += 1
                self._eager_stepped = True
            return loss
        flavour = ("step" if should_step else "accumulate", key)
        if should_step:
            self.optimizer.prepare()            # host half (step count, lr -> device); also allocates its buffers before a capture
        if flavour not in self._graphs:
            self._capture(batch, should_step, key)
        else:
            self._copy_batch(batch, key)
        self._graphs[flavour].replay()
        if should_step:
            if self.distributed and not self._graph_comm():   # one big averaged all-reduce between the two halves of the step
                self.arena.allreduce_all()
                self.optimizer.launch()
                self.arena.zero_()
            self.optimizer_step += 1
        return self._static_loss[flavour]

    def _graph_comm(self):
        """Collectives inside the captured step: only through the direct RCCL C-ABI (csrc/comm.hip), whose launches are plain stream work.
        torch.distributed's gloo (and ProcessGroupNCCL's watchdog) cannot be captured: then the arena is all-reduced in one piece between
        the replayed graph and the optimizer (TSASR_GRAPH_COMM=0 forces that form for A/B runs)."""
        if not (self.distributed and _GRAPH_COMM and getattr(self.arena, "direct", False)):
            return False
        if self._graph_comm_ok is None:   # first question of a multi-rank run (asked by every rank before its first step): probe once
            from . import dp as _dp
            self._graph_comm_ok = _dp.direct_capture_probe(self.device, self.arena.world_size, self.rank, self.arena.group)
            if not self._graph_comm_ok and self.rank == 0:
                import sys
                print("[ts-asr_amd] collectives stay outside the captured step (one all-reduce between replay and optimizer)", file=sys.stderr)
        return self._graph_comm_ok

    def _capture(self, batch, should_step, key):
        if key not in self._static_batches:
            self._static_batches[key] = batch.to(self.device)
        else:
            self._copy_batch(batch, key)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # multi-rank: the RCCL watchdog thread polls events while this thread captures; "thread_local" keeps its (legal) calls from
        # invalidating the capture (single rank keeps the strict default)
        mode = "thread_local" if self.distributed else "global"
        pool = self._graph_pool
        from . import ops as _ops
        # job-table pairs for this graph's batched launches (one per captured flush: per bucket with collectives in the graph, plus the
        # early / duplicate-weight / final flushes), allocated NOW - pinned memory cannot be allocated inside a capture
        n_tables = len(self.arena.buckets) + 6
        _ops.reserve_captured_tables(n_tables)
        self.arena.reserve_captured_tables(n_tables)
        try:
            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                loss, _ = self._device_step(self._static_batches[key], should_step, comm=self._graph_comm())
        except BaseException:
            # a capture that raised half-way: drop the queued launches and their operand references, leave the arena out of "backward"
            _ops.discard_queues()
            self.arena.abort_backward()
            raise
        _ops.upload_captured_tables()             # job tables of the captured flushes: uploaded once, now (replays carry no memcpy node)
        self.arena.upload_captured_tables()
        if self._graph_pool is None:
            self._graph_pool = g.pool()
        flavour = ("step" if should_step else "accumulate", key)
        self._graphs[flavour], self._static_loss[flavour] = g, loss
        self._graph = g

    def _copy_batch(self, batch, key):
        static = self._static_batches[key]
        if batch is static:
            return
        for k in static._keys:
            dst, src = getattr(static, k), getattr(batch, k)
            if isinstance(dst, tuple):
                for d, s_ in zip(dst, src):
                    d.copy_(s_, non_blocking=True)

    def check_gradients(self, loss):
        """Counts non-finite losses on the device (reference: counted, the step is NOT skipped; core.py:1115-1150)."""
        if self._nonfinite_dev is None:
            self._nonfinite_dev = torch.zeros((), dtype=torch.int32, device=loss.device)
        if loss.is_cuda:
            from . import ops as _ops
            _ops.count_nonfinite(loss, self._nonfinite_dev)      # one launch (torch.isfinite alone is five)
        else:
            self._nonfinite_dev.add_((~torch.isfinite(loss.detach())).to(torch.int32).reshape(()))
        return True

    def flush_nonfinite(self):
        """One host read for all the steps since the last flush; raises like the reference when patience is exhausted."""
        if self._nonfinite_dev is not None:
            self.nonfinite_count += int(self._nonfinite_dev.item())
            self._nonfinite_dev.zero_()
        if self.optimizer is not None and hasattr(self.optimizer, "take_skipped_steps"):
            self.skipped_steps += self.optimizer.take_skipped_steps()   # non-finite gradient norm: the fused optimizer left the weights alone
        if "cuda" in self.device:
            from . import ops as _ops
            if _ops.lstm_timeouts():
                raise RuntimeError("persistent LSTM kernel: an inter-workgroup wait timed out (workgroups not co-resident?); "
                                   "set TSASR_LSTM_PERSISTENT=0 to use the per-step kernels")
            from . import rnnt as _rnnt
            if _rnnt.lattice_timeouts():
                raise RuntimeError("RNN-T lattice split over workgroups: a wait for the neighbouring column block timed out (blocks not "
                                   "co-resident?); tsasr_rnnt_lattice_plan(-1, -1, 0) keeps every lattice in one workgroup")
        if self.nonfinite_count > self.nonfinite_patience:
            raise ValueError("Loss is not finite and patience is exhausted.")
        return self.nonfinite_count

    def evaluate_batch(self, batch, stage):
        out = self.compute_forward(batch, stage=stage)
        return self.compute_objectives(out, batch, stage=stage).detach()

    # ---- loops ----------------------------------------------------------------------------------------
    def fit(self, epoch_counter, train_set, valid_set=None, progressbar=None, train_loader_kwargs=None, valid_loader_kwargs=None):
        self.on_fit_start()
        for epoch in epoch_counter:
            self.on_stage_start(Stage.TRAIN, epoch)
            self.modules.train()
            total, n = None, 0
            for batch in train_set:
                self.step += 1
                loss = self.fit_batch(batch)
                total = loss if total is None else total + loss
                n += 1
                if self.nonfinite_flush_every and n % self.nonfinite_flush_every == 0:
                    self.flush_nonfinite()   # raises as the reference does once patience is exceeded (SB/core.py:1115-1150), at most that many steps late
                if self.optimizer_step_limit is not None and self.optimizer_step >= self.optimizer_step_limit:
                    break
            self.flush_nonfinite()
            self.avg_train_loss = float(total / max(n, 1)) if total is not None else 0.0
            self.on_stage_end(Stage.TRAIN, self.avg_train_loss, epoch)
            self.step = 0
            if valid_set is not None:
                self._eval_loop(valid_set, Stage.VALID, epoch)

    def evaluate(self, test_set, max_key=None, min_key=None, progressbar=None, test_loader_kwargs=None):
        return self._eval_loop(test_set, Stage.TEST, None)

    def _eval_loop(self, data, stage, epoch):
        self.on_stage_start(stage, epoch)
        self.modules.eval()
        total, n = None, 0
        with torch.no_grad():
            for batch in data:
                loss = self.evaluate_batch(batch, stage)
                total = loss if total is None else total + loss
                n += 1
        avg = float(total / max(n, 1)) if total is not None else 0.0
        self.on_stage_end(stage, avg, epoch)
        return avg
