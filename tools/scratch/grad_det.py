"""Run-to-run reproducibility of the gradients of a captured accumulate-only graph: prints one checksum line per replay and
parameter group; run the process twice and diff the outputs (weights never change after the warm-up steps)."""
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
for _ in range(4):
    brain.fit_batch(batch)                       # 2-3 eager optimizer steps, then the captured stepping flavour
brain.grad_accumulation_factor = 10 ** 9         # from now on: accumulate-only micro-steps (weights frozen)
names = {id(p): n for n, p in brain.modules.named_parameters()}
groups = {}
for p in brain.arena.params_ordered:
    n = names.get(id(p), "?")
    g = n.split(".")[0] + ("." + n.split(".")[2] if n.startswith(("encoder.layers", "speaker_encoder.layers")) else "")
    if n.startswith("frontend"):
        g = n
    groups.setdefault(g, []).append(p)
for it in range(int(os.environ.get("REPS", "8"))):
    brain.arena.zero_()
    loss = float(brain.fit_batch(batch))
    torch.cuda.synchronize()
    g = brain.arena.grads
    line = [f"rep {it} loss {loss!r}"]
    for name, ps in groups.items():
        hsh = hashlib.md5()
        for p in ps:
            o, n = brain.arena.offset[id(p)], p.numel()
            hsh.update(g[o:o + n].cpu().numpy().tobytes())
        line.append(f"{name}:{hsh.hexdigest()[:6]}")
    print(" ".join(line), flush=True)
    for p in brain.arena.params_ordered:
        if names.get(id(p)) == "frontend.convblock_0.convs.conv_0.conv.bias":
            o, n = brain.arena.offset[id(p)], p.numel()
            v = g[o:o + n].double().cpu()
            print("B1", it, " ".join(f"{x:.9e}" for x in v[:128].tolist()), flush=True)
print("graphs:", len(getattr(brain, "_graphs", {})))
