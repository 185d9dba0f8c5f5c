import importlib, os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
ops = importlib.import_module(bench.PKG + ".ops")
nnet = importlib.import_module(bench.PKG + ".nnet")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
for _ in range(2):
    brain.fit_batch(batch)
seen = collections.Counter()
orig = ops._bf16_weight
def tap(w):
    sh = getattr(w, "_bf16", None)
    if not (sh is not None and getattr(w, "_bf16_ver", -1) == w._version):
        fr = [f for f in traceback.extract_stack()[:-1] if "ts-asr_amd" in f.filename][-3:]
        seen[(tuple(w.shape), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr))] += 1
    return orig(w)
ops._bf16_weight = tap
orig_cd = nnet._cd
seen_cd = collections.Counter()
def tap_cd(x):
    if x.dtype != nnet._COMPUTE_DTYPE:
        fr = [f for f in traceback.extract_stack()[:-1] if "ts-asr_amd" in f.filename][-2:]
        seen_cd[(tuple(x.shape), str(x.dtype), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr))] += 1
    return orig_cd(x)
nnet._cd = tap_cd
conf = importlib.import_module(bench.PKG + ".conformer"); conf._cd = tap_cd
brain.fit_batch(batch)
torch.cuda.synchronize()
print("bf16 weight casts without a shadow:")
for k, v in seen.most_common(): print(v, k)
print("activation casts (_cd):")
for k, v in seen_cd.most_common(): print(v, k)
