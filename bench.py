#!/usr/bin/env python3
"""Headline benchmark: Conformer-Transducer TS-ASR training step (conformer-t_scratch, T=1000 mel frames, B=32/GPU, bf16).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = TSASR.fit_batch on one synthetic batch resident in HBM: speaker branch + front-end + 12-layer Conformer
encoder + predictor + fused joint/head + RNN-T loss, backward, gradient all-reduce (N > 1), clip + AdamW.
Workload = BASELINE.json configs[1]: mel N(0,1) [32,1000,80] (already normalised; fed after A2),
enrollment mel [32,500,80], tokens randint(1,29) [32,120], all lengths 1.0, injection_mode=cat, dropout 0.1 active,
grad_accumulation_factor 1, per-rank data seed 1234+rank, weights random-init (seed 0).
Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "ts-asr_amd"
B_LOCAL, T_MEL, T_ENROLL, U = 32, 1000, 500, 120
# --config: the headline workload (BASELINE.json configs[1] / [2]) and the two other GPU configurations of BASELINE.json
WORKLOADS = {
    "scratch": dict(yaml="conformer-t_scratch_mi355x.yaml", B=32, T=1000, Te=500, U=120, overrides={}, emb=0,
                    name="BASELINE.json configs[1]: conformer-t_scratch 12L d256 (+6L speaker encoder), mel [32,1000,80] + enrollment mel [32,500,80] + tokens [32,120], injection cat, dropout 0.1"),
    "pretrained": dict(yaml="conformer-t_wavlm_mi355x.yaml", B=32, T=1000, Te=500, U=120, overrides={}, emb=512,
                       name="BASELINE.json configs[3]: conformer-t_wavlm (frozen speaker encoder's x-vector given as [32,1,512] -> speaker_proj 512->256 -> cat injection), mel [32,1000,80] + tokens [32,120], dropout 0.1"),
    "none": dict(yaml="conformer-t_none_mi355x.yaml", B=32, T=1000, Te=500, U=120, overrides={}, emb=0,
                 name="train_librispeechmix_none.py / conformer-t_none.yaml: the same transducer WITHOUT a speaker branch (12L d256 encoder, no injection), mel [32,1000,80] + tokens [32,120], dropout 0.1"),
    "longform": dict(yaml="conformer-t_scratch_mi355x.yaml", B=1, T=16000, Te=500, U=1920, emb=0,
                     overrides=dict(causal_encoder=True, frontend_padding="causal"),
                     name="BASELINE.json configs[4]: causal conformer-t (causal encoder + causal front-end padding, as the reference's --causal_encoder True --frontend_padding causal), B=1/GPU, mel [1,16000,80] -> T'=4000, tokens [1,1920], enrollment mel [1,500,80], dropout 0.1"),
    "inject_sum": dict(yaml="conformer-t_scratch_mi355x.yaml", B=32, T=1000, Te=500, U=120, overrides=dict(injection_mode="sum"), emb=0,
                       name="configs[1] with injection_mode: sum (the reference's other mean-pooled injection; pricing run, not the headline)"),
    "inject_xattn": dict(yaml="conformer-t_scratch_mi355x.yaml", B=32, T=1000, Te=500, U=120, overrides=dict(injection_mode="cross_attention"), emb=0,
                         name="configs[1] with injection_mode: cross_attention (the speaker encoder's frames as keys / values of one attention layer in front of the encoder; pricing run, not the headline)"),
    "longform_chunk40": dict(yaml="conformer-t_scratch_mi355x.yaml", B=1, T=16000, Te=500, U=1920, emb=0,
                             overrides=dict(causal_encoder=True, frontend_padding="causal", attention_chunk_size=40),
                             name="BASELINE.json configs[4] with the BUILD EXTENSION chunk=40 (block-causal attention: a frame sees its whole 40-frame chunk and everything before it; the reference has no chunked attention), B=1/GPU, mel [1,16000,80] -> T'=4000, tokens [1,1920]"),
}
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table)


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_brain(device, compute_dtype="bf16", accum=1, overrides=None, yaml_name="conformer-t_scratch_mi355x.yaml"):
    import torch
    hp = importlib.import_module(PKG + ".hparams")
    tsasr = importlib.import_module(PKG + ".recipes.tsasr")
    ov = dict(input_is_feats=True, compute_dtype=compute_dtype, grad_accumulation_factor=accum)
    ov.update(overrides or {})
    with open(os.path.join(ROOT, "hparams", yaml_name)) as f:
        h = hp.load_hyperpyyaml(f, ov)
    run_opts = {"device": device, "compute_dtype": compute_dtype, "grad_accumulation_factor": accum,
                "distributed_launch": int(os.environ.get("WORLD_SIZE", "1")) > 1}
    brain = tsasr.TSASR(h["modules"], h["opt_class"], h, run_opts)
    brain.modules.train()
    return brain, h, torch


# algorithmic HBM bytes per launch of each hand-written kernel family at configs[1] (DESIGN.md "Kernels")
def algorithmic_bytes(B=B_LOCAL, Tp=T_MEL // 4, U1=U + 1, J=640, V=29):
    cells = B * Tp * U1
    act = (B * Tp * J + B * U1 * J) * 2            # enc_proj + dec_proj, bf16
    return {
        "joint_fwd": act + cells * V * 4,                       # read enc/dec, write logits fp32
        "rnnt_loss_fwd": cells * V * 4 + cells * 4 * 4,         # read logits; alpha/beta/lp pairs round trip
        "rnnt_loss_bwd": 2 * cells * V * 4,                     # read logits, write dlogits
        "joint_bwd": cells * V * 4 + act + act,                 # read dlogits + enc/dec, write denc/ddec
    }


def cpu_baseline(brain, torch, budget_steps=6, B=4):   # ~10 s of host work
    """Oracle (CPU restatement of the reference) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import rnnt_ref, tsasr_ref
    batch_mod = importlib.import_module(PKG + ".batch")
    # threads = the CPU share this job is given, not the host's core count: the GPU boxes of the pool hand a one-GPU job 16 CPUs of a
    # 256-thread host (the pool's rule; more threads than the share are time-sliced, not faster). The cgroup quota, when one is set, and
    # the affinity mask are reported next to the figure; TSASR_CPU_BASELINE_THREADS overrides the 16.
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    share = int(os.environ.get("TSASR_CPU_BASELINE_THREADS", "16"))
    cores = max(1, min(len(os.sched_getaffinity(0)), share, int(quota) if quota and quota >= 1 else share))
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()}, affinity={len(os.sched_getaffinity(0))}, cgroup cpu quota={quota})")
    sd = {f"{n}.{k}": v.detach().cpu().float().clone().requires_grad_(v.dtype.is_floating_point)
          for n, m in brain.modules.items() for k, v in m.state_dict().items()}
    cfg = dict(nhead=4, encoder_num_layers=12, speaker_num_layers=6, vocab_size=29, blank_index=0)
    bt = batch_mod.synthetic_batch(B, T_MEL, T_ENROLL, U, feats=True, seed=999)
    cb = {"mixed_feats": bt.mixed_sig.data, "mixed_lens": bt.mixed_sig.lengths, "enroll_feats": bt.enroll_sig.data,
          "enroll_lens": bt.enroll_sig.lengths, "tokens_bos": bt.tokens_bos.data, "tokens_bos_lens": bt.tokens_bos.lengths}
    times = []
    for i in range(1 + budget_steps):
        log(f"cpu baseline step {i}")
        t0 = time.perf_counter()
        logits = tsasr_ref.compute_forward(cb, sd, cfg, "cat", from_feats=True)
        loss = rnnt_ref.transducer_loss_ref_torch(logits, bt.tokens.data, bt.mixed_sig.lengths, bt.tokens.lengths, 0, "mean")
        loss.backward()
        for v in sd.values():
            v.grad = None
        if i > 0:
            times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    # What this "port" figure is worth relative to the REFERENCE: tools/cpu_crosscheck.py times the imported reference modules and this
    # oracle back to back in the build container (identical inputs, weights, exclusions; profiles/r*_cpu_crosscheck.json, committed).
    # reference_ratio = oracle frames/s / reference frames/s there: multiply nothing - it says the port is within that factor of the reference.
    ref_ratio, ref_src = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_cpu_crosscheck.json")), reverse=True):
        try:
            d = json.load(open(path))
            if "reference_ratio" in d:
                ref_ratio, ref_src = d["reference_ratio"], os.path.basename(path)
                break
        except (OSError, ValueError):
            continue
    return {"value": round(B * T_MEL / med, 1), "unit": "frames/s", "cores": cores, "kind": "port", "threads": cores, "cpu_model": cpu_model,
            "reference_ratio": ref_ratio, "reference_ratio_source": ref_src,
            "host": {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota, "cpu_share_of_a_one_gpu_job": share},
            "sample": f"fwd+loss+bwd of the same model/shape at B={B} (T=1000 mel, U=120, 5 s enrollment), fp32, median of {len(times)} steps after 1 warm-up; optimizer step excluded"}


def pmc_traffic(kernel):
    """FETCH_SIZE + WRITE_SIZE bytes per launch of `kernel` (PMC passes of this same bench command, corrected per the gfx950 guide)."""
    import glob
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            for r in json.load(open(path))["kernels"]:
                if r["kernel"] == kernel:
                    return round(r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]), os.path.basename(path)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher (the reference starts its ranks with `python -m torch.distributed.launch
    --nproc_per_node=N`, README.md:38-51): spawn one rank per GPU through torch.distributed.run as a CHILD process and return its exit
    code. Nothing in this (parent) process has touched the GPU - torch.cuda.device_count() does not initialise it - and nothing is
    re-exec'ed. With fewer visible GPUs than ranks the run is a REHEARSAL of the multi-rank code path: every rank on GPU 0, collectives
    over gloo (RCCL refuses two ranks on one device); the JSON line then says so and is not a scaling number."""
    import socket
    import subprocess
    import torch
    env = dict(os.environ)
    have = torch.cuda.device_count()
    if have < n and "TSASR_DIST_BACKEND" not in env:
        env["TSASR_DIST_BACKEND"] = "gloo"
        log(f"{have} GPU(s) visible for {n} ranks: REHEARSAL on one GPU over gloo (not a scaling measurement)")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning: " + " ".join(cmd))
    return subprocess.call(cmd, env=env)


def allreduce_probe(brain, torch, world, reps=5):
    """Measured cost of ONE gradient all-reduce of the whole arena (204 MB fp32 at configs[1]) outside the step, on the path the step uses
    (bare ncclAllReduce of the direct RCCL communicator when there is one, else torch.distributed): ms per call, HIP events."""
    import torch.distributed as dist
    a = brain.arena
    buf = torch.empty_like(a.grads).normal_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    direct = bool(getattr(a, "direct", False))
    capi = importlib.import_module(PKG + "._capi")
    import ctypes

    def once():
        if direct:
            capi.check(capi.lib().tsasr_allreduce_bucket(capi.ptr(buf), buf.numel(), capi.F32, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "tsasr_allreduce_bucket")
        else:
            dist.all_reduce(buf)
    once()
    torch.cuda.synchronize()
    dist.barrier()
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, direct, buf.numel() * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # 2.5 s of timed replays at configs[1]: averages box-to-box jitter, visible to a utilisation sampler
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--ragged", action="store_true", help="lengths U(0.6,1) sorted ascending instead of all 1.0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--config", default="scratch", choices=sorted(WORKLOADS), help="workload (default: the headline, BASELINE.json configs[1])")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch of the workload (pricing runs only: the line is then NOT the headline configuration and says so)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.config])
    if args.batch:
        wl["B"], wl["name"] = args.batch, wl["name"] + f" -- NOT the named configuration: per-GPU batch {args.batch}"
    global B_LOCAL, T_MEL, T_ENROLL, U
    B_LOCAL, T_MEL, T_ENROLL, U = wl["B"], wl["T"], wl["Te"], wl["U"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))     # `python bench.py --gpus N`: this process becomes the launcher, BEFORE anything touches a GPU
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with --gpus N (or run `python bench.py --gpus N`, which spawns them)")
    if os.environ.get("TSASR_DIST_BACKEND") == "gloo":   # rehearsal: every rank on the one visible GPU
        local_rank = 0
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    dp = importlib.import_module(PKG + ".dp")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") always, except for the one-GPU rehearsal of the multi-rank code path (`python bench.py --gpus N` with fewer visible GPUs)
        dp.ddp_init_group({"distributed_launch": True, "distributed_backend": os.environ.get("TSASR_DIST_BACKEND", "nccl")})
    prof = importlib.import_module(PKG + ".prof")
    batch_mod = importlib.import_module(PKG + ".batch")

    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")  # GLUE convs/LSTM: no exhaustive MIOpen search on first use
    log("building model")
    brain, h, _ = build_brain(device, args.dtype, args.accum, wl["overrides"], wl["yaml"])
    log("model built")
    if world > 1:  # identical initial weights on every rank (reference: DDP constructor broadcast)
        for p in brain.modules.parameters():
            torch.distributed.broadcast(p.data, 0)
    batch = batch_mod.synthetic_batch(B_LOCAL, T_MEL, T_ENROLL, U, feats=True, seed=1234 + rank, ragged=args.ragged, enroll_emb_dim=wl["emb"]).to(device)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # fp32 is the parity mode: its GEMMs are the library's exact fp32 path (hipBLASLt), which cannot be stream-captured
    use_graph = (not args.eager) and args.dtype == "bf16"
    if use_graph:
        brain.enable_hip_graph(warmup_steps=min(3, max(2, args.warmup - 1)))
    # graph mode: >= 3 eager steps + the capture step are warm-up; with gradient accumulation both flavours (accumulate-only and
    # stepping micro-batch) must have been captured: one full accumulation cycle runs eagerly first, the second one captures
    n_warm = max(args.warmup, (max(4, 2 * args.accum + 1)) if use_graph else 0)
    for i in range(n_warm):
        brain.fit_batch(batch)
        torch.cuda.synchronize()
        log(f"warm-up step {i + 1} done" + (" (hipGraph captured)" if brain._graph is not None else ""))
    sync()
    graphed = brain._graph is not None
    prof.ENABLED = not graphed          # HIP events cannot bracket kernels inside a replayed graph
    prof.reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = brain.fit_batch(batch)
    sync()
    elapsed = time.perf_counter() - t0
    prof.ENABLED = False
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    loss = loss.clone()
    if graphed:
        # per-kernel durations: the same step, same process, run eagerly right after the timed region with HIP events
        # around every hand-written kernel launch (the graph replays the very same kernels with the same arguments)
        brain._graph_mode = False
        prof.ENABLED = True
        for _ in range(3):
            brain.fit_batch(batch)
        torch.cuda.synchronize()
        prof.ENABLED = False
        brain._graph_mode = True
    feat_ms, aug_ms = None, {}
    if rank == 0:   # Fbank + sentence norm are timed separately, on wav ~ N(0, 0.1^2) [32, 159840] (the step is fed features)
        wav = torch.randn(B_LOCAL, T_MEL * 160 - 160, device=device) * 0.1
        wlens = torch.ones(B_LOCAL, device=device)
        fx, nm = brain.modules.feature_extractor, brain.modules.normalizer
        with torch.no_grad():
            for _ in range(3):
                nm(fx(wav), wlens, epoch=0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                nm(fx(wav), wlens, epoch=0)
            e1.record()
            torch.cuda.synchronize()
            feat_ms = e0.elapsed_time(e1) / 10
            # the two augmenters of compute_forward (`augment: True`), same shapes: SpecAugment on [32,1000,80] features (draw + warp +
            # masks), speed perturbation of the [32,159840] waveform to 95 % (the headline step runs with the YAML default augment: False)
            feats = nm(fx(wav), wlens, epoch=0)
            aug_ms = {}
            for name, fn in (("specaugment_ms", lambda: brain.modules.augmentation(feats)),
                             ("speed_perturb_ms", lambda: brain.modules.speed_perturb.resamplers[0](wav))):
                for _ in range(3):
                    fn()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                aug_ms[name] = round(e0.elapsed_time(e1) / 10, 4)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    comm_info = {}
    if world > 1:
        ar_ms, direct, nbytes = allreduce_probe(brain, torch, world)
        dpm = importlib.import_module(PKG + ".dp")
        comm_info = {"backend": torch.distributed.get_backend(), "rccl_direct_ranks": int(dpm._DIRECT["ranks"]), "collectives_in_graph": bool(brain._graph_comm()) if brain._graph is not None else False,
                     "allreduce_ms_whole_arena": round(ar_ms, 3), "allreduce_bytes": nbytes,
                     "allreduce_busbw_GBps": round(2 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9, 1),
                     "rehearsal_on_one_gpu": os.environ.get("TSASR_DIST_BACKEND") == "gloo"}
    nonfinite = brain.flush_nonfinite()
    # per-launch durations: mean of the HIP-event brackets of every instrumented launch, NET of the empty event-pair time measured in
    # this process (prof.event_overhead_ms x 0.8, ~3 us: the bracket's own two timestamp packets) - without it the brackets read 17 us
    # where rocprofv3 (profiles/r04_kernel_trace.md) reads 11.7 us for the same launches; no fastest-of-N
    kern = prof.collect(subtract_overhead=True, repeats=3 if graphed else args.steps, fastest=False)

    if rank == 0:
        frames = world * B_LOCAL * T_MEL * args.steps
        ms_step = elapsed / args.steps * 1e3
        ab = algorithmic_bytes(B_LOCAL, T_MEL // 4, U + 1)
        wk = prof.work()
        wb = prof.algorithmic_bytes()
        fam = {}
        for k, (n, ms) in kern.items():
            e = {"launches": n, "avg_ms": round(ms, 4)}
            if k in ab and ms > 0:
                e["algorithmic_GBps"] = round(ab[k] / (ms * 1e-3) / 1e9, 1)
            if wk.get(k, 0) > 0 and ms > 0:
                e["avg_GFLOP_per_launch"] = round(wk[k] / n / 1e9, 3)
                e["TFLOPps"] = round(wk[k] / n / (ms * 1e-3) / 1e12, 1)
            if wb.get(k, 0) > 0:
                e["algorithmic_MB_per_launch"] = round(wb[k] / n / 1e6, 1)
            fam[k] = e
        # dominant hand-written kernel = largest share of the step among the instrumented launches
        dom = max(fam, key=lambda k: fam[k]["avg_ms"] * fam[k]["launches"], default=None)
        latency_bound = None
        if dom is not None and "TFLOPps" not in fam[dom] and dom not in ab:
            # (long-form, B = 1: the largest launch is a chain of dependent steps - persistent LSTM / RNN-T lattice - priced by steps, not by
            # flops or bytes: the roofline object then describes the largest kernel that has a flop or byte model, and says so)
            latency_bound = dom
            cands = [k for k in fam if "TFLOPps" in fam[k] or k in ab]
            dom = max(cands, key=lambda k: fam[k]["avg_ms"] * fam[k]["launches"], default=None)
        roof = None
        if dom is not None and "TFLOPps" in fam[dom]:
            roof = {"kernel": dom, "bound": "mfma", "achieved": fam[dom]["TFLOPps"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(fam[dom]["TFLOPps"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                    "algorithmic_flops_per_launch": fam[dom]["avg_GFLOP_per_launch"] * 1e9, "avg_launch_ms": fam[dom]["avg_ms"],
                    "note": "dominant kernel = the rocprofv3 kernel name (template instantiation) with the largest share of the step among the hand-written launches; achieved = sum of 2*M*N*K of its launches / sum of their HIP-event durations on the launch stream (mean over 3 instrumented eager steps of this process, net of the measured empty event-pair time: event_pair_overhead_ms)", "event_pair_overhead_ms": round(prof.out_overhead[0], 5)}
        elif dom is not None and dom in ab:
            ach = ab[dom] / (fam[dom]["avg_ms"] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": fam[dom]["avg_ms"]}
        if roof is not None:   # HBM-side bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_summary.py), newest round
            roof["traffic"], roof["traffic_source"] = pmc_traffic(roof["kernel"])
            if latency_bound is not None:
                roof["largest_launch_without_flop_or_byte_model"] = {"kernel": latency_bound, **fam[latency_bound]}
        rnnt_ms = sum(fam[k]["avg_ms"] for k in ("joint_fwd", "rnnt_loss_fwd", "rnnt_loss_bwd", "joint_bwd") if k in fam)
        out = {
            "metric": f"utterance-frames/sec ({ {'pretrained': 'conformer-t_wavlm', 'none': 'conformer-t_none'}.get(args.config, 'conformer-t_scratch') } training step, T={T_MEL}, B={B_LOCAL}/GPU)",
            "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["name"] + ", lens " + ("U(0.6,1) ascending" if args.ragged else "1.0"),
                       "global_batch": world * B_LOCAL, "grad_accumulation_factor": args.accum, "parallelism": f"dp{world}",
                       "hip_graph": brain._graph is not None},
            "frames_per_sec_per_gpu": round(frames / elapsed / world, 1),
            "rnnt_joint_loss_ms": round(rnnt_ms, 4),
            "fbank_sentnorm_ms": None if feat_ms is None else round(feat_ms, 4),
            **aug_ms,
            "loss": round(float(loss), 4), "nonfinite_steps": nonfinite,
            "hip_kernels": fam,
            "roofline": roof,
        }
        if comm_info:
            out["comm"] = comm_info
        if not args.no_cpu_baseline and world == 1 and args.config == "scratch":
            out["cpu_baseline"] = cpu_baseline(brain, torch)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
