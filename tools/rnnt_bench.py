"""Lab bench of the RNN-T loss kernels (rnnt_lp + rnnt_alphabeta, rnnt_grad) on their own, per lattice plan.

    python tools/rnnt_bench.py [--long] [--reps 20]

    python tools/rnnt_bench.py --long --plans         # every lattice plan of csrc/rnnt.hip rnnt_plan (tsasr_rnnt_lattice_plan) in turn

The costs and the gradient norm are printed so that the plans can be compared bit for bit.
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnnt = importlib.import_module("ts-asr_amd.rnnt")
C = importlib.import_module("ts-asr_amd._capi")

PLANS = [("default", (-1, -1, -1)), ("frame-major, 4 waves (round 3)", (0, 4, 0)), ("skewed, 4 waves", (1, 4, 0)), ("skewed, 8 waves", (1, 8, 0)),
         ("skewed, 16 waves", (1, 16, 0)), ("split over workgroups, 1 column per thread", (1, 8, 1)), ("split, 2 columns", (1, 8, 2)),
         ("split, 4 columns", (1, 8, 4))]


def run(B, T, U, V, reps, ragged):
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(5)
    logits = torch.randn(B, T, U + 1, V, generator=g).to(dev).requires_grad_(True)
    targets = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32).to(dev)
    tl = torch.full((B,), T, dtype=torch.int32)
    ul = torch.full((B,), U, dtype=torch.int32)
    if ragged and B > 1:
        tl = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32); tl[0] = T
        ul = torch.randint(U // 2, U + 1, (B,), generator=g, dtype=torch.int32); ul[0] = U
    tl, ul = tl.to(dev), ul.to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fw, bw = [], []
    for r in range(reps + 3):
        logits.grad = None
        ev[0].record()
        costs = rnnt.rnnt_costs(logits, targets, tl, ul, 0)
        ev[1].record()
        costs.sum().backward()
        ev[2].record()
        torch.cuda.synchronize()
        if r >= 3:
            fw.append(ev[0].elapsed_time(ev[1]) * 1e3)
            bw.append(ev[1].elapsed_time(ev[2]) * 1e3)
    fw.sort(); bw.sort()
    print(f"B={B} T={T} U={U} ragged={ragged}: loss forward (lp + alpha/beta) {fw[len(fw) // 2]:9.1f} us   backward (grad + sum) {bw[len(bw) // 2]:9.1f} us   "
          f"costs[0]={costs[0].item():.6f} sum={costs.double().sum().item():.6f} |grad|={logits.grad.double().norm().item():.9f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true", help="configs[4]: B=1, T'=4000, U=1920 (default: configs[1]: B=32, T'=250, U=120)")
    ap.add_argument("--mid", action="store_true", help="B=8, T'=1000, U=400")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--plans", action="store_true", help="every lattice plan in turn instead of the default one")
    a = ap.parse_args()
    for name, plan in (PLANS if a.plans else PLANS[:1]):
        C.lib().tsasr_rnnt_lattice_plan(*plan)
        print(f"-- lattice plan: {name}")
        if a.long:
            run(1, 4000, 1920, 29, a.reps, False)
        elif a.mid:
            run(8, 1000, 400, 29, a.reps, True)
        else:
            run(32, 250, 120, 29, a.reps, False)
            run(32, 250, 120, 29, a.reps, True)
    C.lib().tsasr_rnnt_lattice_plan(-1, -1, -1)
