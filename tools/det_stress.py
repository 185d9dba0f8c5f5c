#!/usr/bin/env python3
"""Determinism stress of the training step (configs[0] model): R repetitions of {eager run, hipGraph run} on fresh brains, batches of
two shapes alternating (the length-bucketed case of tests/test_model_gpu.py::test_hip_graph_cache_per_batch_shape), optionally with
gradient accumulation. Every run is compared with repetition 0's EAGER run: losses bit for bit and - with --snap 1 - every parameter and
its first Adam moment after every step (the moment is (1 - b1) x gradient + ..., so the first differing moment names the first gradient
that went wrong). Prints one line per run that differs; exit code 1 when any did.
usage: python tools/det_stress.py [--reps 20] [--steps 12] [--accum 1] [--shapes 2] [--snap 1] [--modes eager,graph]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

entry = importlib.import_module("__graft_entry__")
from oracle.golden_recipe import golden_inputs  # noqa: E402
from test_model_gpu import make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--accum", type=int, default=1)
ap.add_argument("--shapes", type=int, default=2)
ap.add_argument("--snap", type=int, default=1)
ap.add_argument("--modes", default="eager,graph")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--poison", type=int, default=0, help="graph mode: NaN into every inactive block of the captured step's memory pool before every step")
ap.add_argument("--fbank", type=int, default=0, help="keep the mixture Fbank's input checksum and full output of every step (persistent buffers, written inside the graph) and compare element by element")
ap.add_argument("--probe", type=int, default=0, help="forward hooks: checksum of every module output into a persistent device vector (inside the graph)")
args = ap.parse_args()
DEV = "cuda"
inp = golden_inputs()
cuts = [1.0, 0.75, 0.5]
variants = [{k: (v[:, : int(v.shape[1] * c)] if k in ("mixed_sig", "enroll_sig") else v) for k, v in inp.items()} for c in cuts[: args.shapes]]


def snapshot(brain):
    out = {}
    a, opt = brain.arena, brain.optimizer
    for n, p in brain.modules.named_parameters():
        o = a.offset.get(id(p))
        if o is None:
            continue
        m = opt.exp_avg[o:o + p.numel()] if hasattr(opt, "exp_avg") else p.data.reshape(-1)
        out[n] = (p.data.detach().reshape(-1).cpu(), m.detach().cpu())
    return out


def add_probes(brain):
    """Checksums (float64 sum and sum of squares) of every sub-module's output, written by reduction kernels straight into a persistent
    vector (allocated before any capture): compared run against run they name the first module whose forward output went wrong."""
    names, mods = [], []
    for n, m in brain.modules.named_modules():
        if n and len(list(m.children())) == 0 or n.count(".") <= 2 and n:
            names.append(n)
            mods.append(m)
    vec = torch.zeros(2 * len(names), dtype=torch.float64, device=DEV)

    def mk(i):
        def hook(mod, inp, out):
            t = out[0] if isinstance(out, (tuple, list)) else out
            if not torch.is_tensor(t) or not t.is_cuda or t.numel() == 0:
                return
            with torch.no_grad():
                d = t.detach().reshape(1, -1).double()
                torch.sum(d, dim=1, out=vec[2 * i:2 * i + 1])
                torch.sum(d * d, dim=1, out=vec[2 * i + 1:2 * i + 2])
        return hook
    for i, m in enumerate(mods):
        m.register_forward_hook(mk(i))
    return names, vec


def poison_pool():
    """NaN (0x7FC07FC0: a NaN as fp32 and as two bf16) into every inactive block of every private (graph) memory pool."""
    import ctypes
    capi = importlib.import_module("ts-asr_amd._capi")
    n = 0
    for seg in torch.cuda.memory_snapshot():
        if tuple(seg.get("segment_pool_id", (0, 0))) == (0, 0):
            continue
        addr = seg["address"]
        for b in seg["blocks"]:
            a = b.get("address", addr)
            if b["state"] == "inactive" and b["size"] >= 4:
                assert capi.lab().tsasr_lab_fill(ctypes.c_void_p(a), 0x7FC07FC0, b["size"] // 4, capi.stream_ptr()) == 0
                n += b["size"]
            addr += b["size"]
    return n


def add_fbank_taps(brain):
    """Mixture Fbank: float64 checksum of the input waveform (pre-hook) and a full copy of the output (hook), per batch shape, written by
    kernels inside the captured step into buffers allocated before any capture."""
    fx = brain.modules.feature_extractor
    taps = {"in": torch.zeros(4, dtype=torch.float64, device=DEV), "out": {}, "last": None}
    for v in variants:
        L = v["mixed_sig"].shape[1]
        taps["out"][L] = torch.zeros(v["mixed_sig"].shape[0], 1 + L // 160, 80, device=DEV)

    def pre(mod, inp):
        w = inp[0]
        with torch.no_grad():
            torch.sum(w.detach().reshape(1, -1).double(), dim=1, out=taps["in"][0:1])
            torch.sum(mod.window.reshape(1, -1).double(), dim=1, out=taps["in"][1:2])
            torch.sum(mod.fbank_matrix.reshape(1, -1).double(), dim=1, out=taps["in"][2:3])

    def post(mod, inp, out):
        with torch.no_grad():
            torch.add(out.detach().float(), 0.0, out=taps["out"][inp[0].shape[1]])
        taps["last"] = inp[0].shape[1]
    fx.register_forward_pre_hook(pre)
    fx.register_forward_hook(post)
    return taps


def run(mode):
    brain, h = entry._config1_brain(DEV, args.dtype)
    taps = add_fbank_taps(brain) if args.fbank else None
    probe_names, probe_vec = add_probes(brain) if args.probe else ([], None)
    brain.grad_accumulation_factor = args.accum
    brain.modules.train()
    if mode == "graph":
        brain.enable_hip_graph(warmup_steps=2)
    batches = [make_batch(v).to(DEV) for v in variants]
    losses, snaps, probes, fb_log = [], [], [], []
    for i in range(args.steps):
        if args.poison and mode == "graph" and brain._graphs:
            run.poisoned = poison_pool()
        losses.append(float(brain.fit_batch(batches[i % len(batches)])))
        if args.snap:
            snaps.append(snapshot(brain))
        if args.probe:
            probes.append(probe_vec.cpu().clone())
        if args.fbank:
            L = variants[i % len(batches)]["mixed_sig"].shape[1]
            fb_log.append((taps["in"].cpu().clone(), taps["out"][L].cpu().clone()))
    torch.cuda.synchronize()
    run.fb_log = fb_log
    run.probe_names = probe_names
    return losses, (snaps, probes)


ref_losses, (ref_snaps, ref_probes) = run("eager")
ref_fb = run.fb_log
print(f"reference (eager): {[round(x, 5) for x in ref_losses]}", flush=True)
bad = 0
total = 0
for r in range(args.reps):
    for mode in args.modes.split(","):
        losses, (snaps, probes) = run(mode)
        total += 1
        dl = [i for i, (a, b) in enumerate(zip(losses, ref_losses)) if a != b]
        ds, detail = None, ""
        if args.snap:
            for i, (s, t) in enumerate(zip(snaps, ref_snaps)):
                names = []
                for n in t:
                    pe, me = t[n]
                    pg, mg = s[n]
                    if not torch.equal(me, mg) or not torch.equal(pe, pg):
                        nm = int((me != mg).sum())
                        names.append(f"{n}[{nm}/{me.numel()} m {float((me - mg).abs().max()):.2e} rel {float((me - mg).abs().max() / (me.abs().max() + 1e-30)):.1e}]")
                if names:
                    ds, detail = i, f"{len(names)}/{len(t)} tensors: " + "; ".join(names[:10])
                    break
        pdetail = ""
        if args.probe:
            for i, (a, b) in enumerate(zip(probes, ref_probes)):
                diff = [k for k in range(len(run.probe_names)) if not torch.equal(a[2 * k:2 * k + 2], b[2 * k:2 * k + 2])]
                if diff:
                    pdetail = f"; first probe difference at step {i}: {len(diff)}/{len(run.probe_names)} modules: " + ", ".join(
                        f"{run.probe_names[k]}({float((a[2 * k + 1] - b[2 * k + 1]).abs() / (b[2 * k + 1].abs() + 1e-300)):.1e})" for k in diff[:14])
                    break
        if args.fbank:
            for i, ((ci, co), (ri, ro)) in enumerate(zip(run.fb_log, ref_fb)):
                if not torch.equal(ci, ri) or not torch.equal(co, ro):
                    d = (co != ro).nonzero()
                    pdetail += (f"; FBANK step {i}: input checksums equal: {torch.equal(ci, ri)} ({ci.tolist()} vs {ri.tolist()}); output differs in {d.shape[0]} of {co.numel()}"
                                f" values, utterances {sorted(set(d[:, 0].tolist()))}, frames {sorted(set(d[:, 1].tolist()))[:12]}, bins {sorted(set(d[:, 2].tolist()))[:12]}; "
                                + ", ".join(f"{tuple(k.tolist())}: {float(co[tuple(k.tolist())])!r} vs {float(ro[tuple(k.tolist())])!r}" for k in d[:6]))
                    later_in = [j for j, ((cj, _), (rj, _)) in enumerate(zip(run.fb_log, ref_fb)) if not torch.equal(cj, rj)]
                    later_out = [j for j, ((_, oj), (_, qj)) in enumerate(zip(run.fb_log, ref_fb)) if not torch.equal(oj, qj)]
                    pdetail += f"; steps whose Fbank INPUT checksums differ: {later_in}; steps whose Fbank OUTPUT differs: {later_out}"
                    break
        if dl or ds is not None or pdetail:
            bad += 1
            print(f"rep {r} {mode}: losses differ at steps {dl} (max rel {max((abs(losses[i] - ref_losses[i]) / abs(ref_losses[i]) for i in dl), default=0):.2e});"
                  f" first state difference after step {ds}: {detail[:600]}{pdetail}", flush=True)
    if r % 5 == 4:
        print(f"... {r + 1} reps, {bad} of {total} runs differ", flush=True)
if args.poison:
    print(f"poisoned {getattr(run, 'poisoned', 0) / 1e6:.1f} MB of pool memory before the last step")
print(f"RESULT {bad} of {total} runs differ from the reference (accum {args.accum}, shapes {args.shapes}, snap {args.snap}, modes {args.modes}, "
      f"env {[k + '=' + v for k, v in os.environ.items() if k.startswith('TSASR_') or k.startswith('PYTORCH_')]})", flush=True)
sys.exit(1 if bad else 0)
