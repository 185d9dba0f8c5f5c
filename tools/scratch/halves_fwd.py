"""Pricing run: the mixture encoder's forward on the whole batch vs two half-batches on two streams, each captured in a hipGraph (scratch)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = "cuda:0"
brain, h, _ = bench.build_brain(dev)
batch_mod = importlib.import_module("ts-asr_amd.batch")
bt = batch_mod.synthetic_batch(32, 1000, 500, 120, feats=True, seed=1).to(dev)
m = brain.modules
with torch.no_grad():
    mel = bt.mixed_sig.data.to(torch.bfloat16) if bt.mixed_sig.data.dtype != torch.bfloat16 else bt.mixed_sig.data
    feats = m.frontend(mel)
    lens = bt.mixed_sig.lengths
    spk = (torch.randn(32, 1, 256, device=dev) * 0.1).to(feats.dtype)
    elens = torch.ones(32, device=dev)
    print("encoder input", tuple(feats.shape), feats.dtype)
    def full():
        return m.encoder(feats, lens, spk, elens)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    fa, fb, la, lb, sa, sb, ea, eb = feats[:16].contiguous(), feats[16:].contiguous(), lens[:16].contiguous(), lens[16:].contiguous(), spk[:16].contiguous(), spk[16:].contiguous(), elens[:16].contiguous(), elens[16:].contiguous()
    def halves():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ya = m.encoder(fa, la, sa, ea)
        with torch.cuda.stream(s2):
            yb = m.encoder(fb, lb, sb, eb)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return ya, yb
    def one_half():
        return m.encoder(fa, la, sa, ea)
    res = {}
    for name, fn in (("full batch", full), ("two halves, two streams", halves), ("one half alone", one_half)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:28s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per forward (12 layers, hipGraph replay)")
