"""Which parameter gradients go through autograd's AccumulateGrad (an ATen add each) instead of the arena's batched add (scratch)."""
import importlib, sys
sys.path.insert(0, ".")
import torch
import bench
wl = dict(bench.WORKLOADS["scratch"])
bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U = wl["B"], wl["T"], wl["Te"], wl["U"]
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, wl["overrides"], wl["yaml"])
batch_mod = importlib.import_module(bench.PKG + ".batch")
ops = importlib.import_module(bench.PKG + ".ops")
batch = batch_mod.synthetic_batch(wl["B"], wl["T"], wl["Te"], wl["U"], feats=True, seed=1234, ragged=False, enroll_emb_dim=wl["emb"]).to("cuda:0")
for _ in range(2):
    brain.fit_batch(batch)
names = {id(p): n for n, p in brain.modules.named_parameters()}
orig = ops._pgrad
def logged(param, g, shape=None):
    r = orig(param, g, shape)
    if r is not None:
        print("AccumulateGrad:", names.get(id(param), "?"), tuple(param.shape), "g", tuple(g.shape), g.dtype, "contig", g.is_contiguous())
    return r
ops._pgrad = logged
hooks = []
for n, p in brain.modules.named_parameters():
    pass
brain.fit_batch(batch)
torch.cuda.synchronize()
# parameters whose .grad is written by autograd directly (no _pgrad at all): count AccumulateGrad nodes via hooks
cnt = {}
def mk(n):
    def hook(g):
        cnt[n] = cnt.get(n, 0) + 1
        return None
    return hook
hs = [p.register_hook(mk(n)) for n, p in brain.modules.named_parameters() if p.requires_grad]
brain.fit_batch(batch)
torch.cuda.synchronize()
print("parameters that received a gradient THROUGH autograd (tensor hook fired):")
for n, c in cnt.items():
    print("  ", n, c)
