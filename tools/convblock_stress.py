#!/usr/bin/env python3
"""Run-to-run bit stability of the fused ConvBlock kernels with a second hardware queue busy (round 1 met a packed-fp32 instruction
whose low bits depended on what else was in flight): N repeats of forward + backward on fixed inputs, every output and parameter
gradient compared bitwise with the first repeat, while another stream runs GEMMs. usage: python tools/convblock_stress.py [repeats]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("ts-asr_amd.ops")
DEV = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200

def params(cin, fo, co=128):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(DEV)  # noqa: E731
    return dict(w1=r(co, cin, 3, 3, scale=0.3 / cin ** 0.5), b1=r(co, scale=0.1), w2=r(co, cin, 1, 1, scale=1 / cin ** 0.5), b2=r(co, scale=0.1),
                g1=1 + r(fo, co, scale=0.1), be1=r(fo, co, scale=0.1), g2=1 + r(fo, co, scale=0.1), be2=r(fo, co, scale=0.1))

def run(x, P, p):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xg = x.clone().requires_grad_(x.shape[-1] > 1)
    conv_params, ln_params = (Pg["w1"], Pg["b1"], Pg["w2"], Pg["b2"]), (Pg["g1"], Pg["be1"], Pg["g2"], Pg["be2"])
    if x.shape[-1] == 1:
        out = ops._FrontendBlockFn.apply(xg.squeeze(-1), None, None, conv_params, ln_params, 0, 0, False, 0.01, 1e-5, p, 11, p, 12, *conv_params, *ln_params)
    else:
        y1, y2 = ops._FrontendConvFn.apply(xg, Pg["w1"], Pg["b1"], Pg["w2"], Pg["b2"], False)
        out = ops._FrontendBlockFn.apply(None, y1, y2, None, ln_params, x.shape[1], x.shape[2], False, 0.01, 1e-5, p, 11, p, 12, None, None, None, None, *ln_params)
    out.backward(torch.ones_like(out))
    ops.reduce_flush()
    res = {"out": out.detach()}
    res.update({k: v.grad for k, v in Pg.items()})
    if xg.grad is not None:
        res["dx"] = xg.grad
    return res

side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
bad_total = 0
for name, shape, cin in [("block1 B=4 T=200", (4, 200, 80, 1), 1), ("block1 B=32 T=1000", (32, 1000, 80, 1), 1), ("block2 B=4 T'=100", (4, 100, 40, 128), 128),
                         ("block2 B=32 T'=500", (32, 500, 40, 128), 128)]:
    x = torch.randn(*shape, device=DEV).bfloat16()
    P = params(cin, (shape[2] - 1) // 2 + 1)
    for p in (0.0, 0.1):
        ref = run(x, P, p)
        torch.cuda.synchronize()
        bad = {}
        for it in range(reps):
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
            got = run(x, P, p)
            torch.cuda.synchronize()
            for k in ref:
                if not torch.equal(ref[k], got[k]):
                    bad[k] = bad.get(k, 0) + 1
        bad_total += len(bad)
        print(f"{name} p={p}: {reps} repeats, mismatching tensors: {bad if bad else 'none'}")
sys.exit(1 if bad_total else 0)
