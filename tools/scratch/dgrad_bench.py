"""Front-end block 2 data gradient: gathered-GEMM kernel vs dy.Wm + col2im, HIP-event timed (scratch)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("ts-asr_amd.ops")
dev = "cuda"
for (B, T, F, Ci) in ((32, 500, 40, 64), (32, 250, 40, 64)):
    Co = 128
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, F, Ci, generator=g).to(torch.bfloat16).to(dev)
    w1 = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(dev); w2 = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).to(dev)
    b1, b2 = torch.randn(Co).to(dev), torch.randn(Co).to(dev)
    To, Fo = (T - 1) // 2 + 1, (F - 1) // 2 + 1
    d1 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16).to(dev); d2 = torch.randn(B, To, Fo, Co, generator=g).to(torch.bfloat16).to(dev)
    out = {}
    for mode in (True, False):
        ops.CONV_DGRAD_IMPLICIT = mode
        xl = x.clone().requires_grad_()
        y1, y2 = ops._FrontendConvFn.apply(xl, w1, b1, w2, b2, False)
        ts = []
        for it in range(12):
            xl.grad = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.autograd.backward([y1, y2], [d1, d2], retain_graph=True, inputs=[xl]); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out[mode] = xl.grad.float()
        print(f"B={B} T={T} F={F} Ci={Ci} dgrad_implicit={mode}: backward wrt x only (incl. wgrad launches? no: inputs=[x]) median {sorted(ts)[6]*1e3:.1f} us")
    print("  rel diff", float((out[True] - out[False]).norm() / out[False].norm()))
