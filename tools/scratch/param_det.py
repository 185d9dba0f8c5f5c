"""Run-to-run reproducibility of the WEIGHTS after every optimizer step of the default (graph, forked branch) training loop:
one checksum per parameter group and step; run the process several times and diff."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
brain.modules.train()
if os.environ.get("GRAPH", "1") == "1":
    brain.enable_hip_graph(warmup_steps=2)
named = list(brain.modules.named_parameters())
def grp(n):
    t = n.split(".")
    if n.startswith(("encoder.layers", "speaker_encoder.layers")):
        return t[0] + "." + t[2]
    if n.startswith(("frontend", "speaker_frontend")):
        return n
    return t[0]
for it in range(int(os.environ.get("REPS", "10"))):
    loss = float(brain.fit_batch(batch))
    torch.cuda.synchronize()
    hs = {}
    for n, p in named:
        hs.setdefault(grp(n), hashlib.md5()).update(p.detach().float().cpu().numpy().tobytes())
    print(f"rep {it} loss {loss!r} " + " ".join(f"{k}:{v.hexdigest()[:6]}" for k, v in hs.items()), flush=True)
