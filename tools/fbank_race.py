#!/usr/bin/env python3
"""Is the log-mel front-end (csrc/features.hip) stable when two of them run side by side on two HIP streams - as the training step runs
the mixture's and the enrollment's (recipes/tsasr.py forks the speaker branch)? Serial reference against N concurrent eager runs and N
replays of a captured two-branch graph; reports every mismatch with where it sits (utterance, frame, mel bin, the two values).
usage: python tools/fbank_race.py [iters]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

nnet = importlib.import_module("ts-asr_amd.nnet")
from oracle.golden_recipe import golden_inputs  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = "cuda"
inp = golden_inputs()
mix = torch.from_numpy(inp["mixed_sig"]).to(dev)
enr = torch.from_numpy(inp["enroll_sig"]).to(dev)
ml, el = torch.from_numpy(inp["mixed_lens"]).to(dev), torch.from_numpy(inp["enroll_lens"]).to(dev)
fb = nnet.Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).to(dev)
nm = nnet.InputNormalization(norm_type="sentence")


def both(sa, sb):
    with torch.cuda.stream(sa):
        fa = fb(mix)
        na = nm(fa, ml)
    with torch.cuda.stream(sb):
        fe = fb(enr)
        ne = nm(fe, el)
    return fa, na, fe, ne


cur = torch.cuda.current_stream()
ref = [t.clone() for t in both(cur, cur)]
torch.cuda.synchronize()
names = ["fbank(mix)", "norm(mix)", "fbank(enroll)", "norm(enroll)"]


def report(tag, outs, it):
    bad = 0
    for n, o, r in zip(names, outs, ref):
        if not torch.equal(o, r):
            bad += 1
            d = (o != r).nonzero()
            k = tuple(d[0].tolist())
            print(f"  {tag} iter {it}: {n} differs in {d.shape[0]} of {o.numel()} values; first at {k}: {float(o[k])!r} vs {float(r[k])!r};"
                  f" utterances {sorted(set(d[:, 0].tolist()))}, frames {int(d[:, 1].min())}..{int(d[:, 1].max())}", flush=True)
    return bad


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for label, sync_each in (("eager two streams, sync each", True), ("eager two streams, free running", False)):
    nbad, keep = 0, []
    for it in range(iters):
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        outs = both(s1, s2)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        for t in outs:
            t.record_stream(cur)
        if sync_each:
            torch.cuda.synchronize()
            nbad += 1 if report(label, outs, it) else 0
        else:
            keep.append([t.clone() for t in outs])
            if len(keep) == 50 or it == iters - 1:
                torch.cuda.synchronize()
                for j, o in enumerate(keep):
                    nbad += 1 if report(label, o, it - len(keep) + 1 + j) else 0
                keep = []
    print(f"{label}: {nbad} of {iters} iterations differ", flush=True)

# captured: fork inside one graph
g = torch.cuda.CUDAGraph()
both(s1, s2)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    c = torch.cuda.current_stream()
    s1.wait_stream(c)
    s2.wait_stream(c)
    gouts = both(s1, s2)
    c.wait_stream(s1)
    c.wait_stream(s2)
nbad = 0
for it in range(iters):
    g.replay()
    torch.cuda.synchronize()
    nbad += 1 if report("graph replay", gouts, it) else 0
print(f"graph replay (two forked branches): {nbad} of {iters} iterations differ", flush=True)
