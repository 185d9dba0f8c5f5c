import importlib, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
for _ in range(2):
    brain.fit_batch(batch)
seen = []
names = {id(p): n for n, p in brain.modules.named_parameters()}
hs = [p.register_post_accumulate_grad_hook(lambda p: seen.append(names[id(p)])) for p in brain.modules.parameters() if p.requires_grad]
brain.fit_batch(batch)
torch.cuda.synchronize()
import re
c = collections.Counter(re.sub(r"\.\d+\.", ".N.", n) for n in seen)
print(len(seen), "of", sum(1 for p in brain.modules.parameters() if p.requires_grad))
c2 = collections.Counter(n.split(".")[0] + ":" + n.split(".")[-1] + ":" + str(tuple(dict(brain.modules.named_parameters())[n].shape)) for n in seen)
for k, v in sorted(c2.items()): print("  ", v, k)
sys.exit(0)
print(len(seen), "parameters received their gradient through AccumulateGrad:")
for k, v in c.most_common(): print(v, k)
