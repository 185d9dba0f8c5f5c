"""Topology of the captured step graph: nodes, edges, forks (nodes with more than one successor), joins (scratch). usage: graph_shape.py <config> [dummy]"""
import importlib, os, re, sys, collections
sys.path.insert(0, ".")
import torch
cfg = sys.argv[1] if len(sys.argv) > 1 else "none"
import bench
graphs = []
_Orig = torch.cuda.CUDAGraph
class Dbg(_Orig):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        return g
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.enable_debug_mode()
        graphs.append(self)
torch.cuda.CUDAGraph = Dbg
wl = dict(bench.WORKLOADS[cfg])
bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U = wl["B"], wl["T"], wl["Te"], wl["U"]
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1, wl["overrides"], wl["yaml"])
batch_mod = importlib.import_module(bench.PKG + ".batch")
batch = batch_mod.synthetic_batch(wl["B"], wl["T"], wl["Te"], wl["U"], feats=True, seed=1234, ragged=False, enroll_emb_dim=wl["emb"]).to("cuda:0")
brain.enable_hip_graph(warmup_steps=3)
for _ in range(5):
    brain.fit_batch(batch)
torch.cuda.synchronize()
for i, g in enumerate(graphs):
    path = f"/tmp/graph_{i}.dot"
    g.debug_dump(path)
    txt = open(path).read()
    edges = re.findall(r'"?([\w\.\-:]+)"?\s*->\s*"?([\w\.\-:]+)"?', txt)
    out, inn = collections.Counter(a for a, b in edges), collections.Counter(b for a, b in edges)
    nodes = set(out) | set(inn)
    print(f"graph {i}: {len(nodes)} nodes, {len(edges)} edges, forks (out-degree > 1): {sum(1 for n in nodes if out[n] > 1)}, joins (in-degree > 1): {sum(1 for n in nodes if inn[n] > 1)}, roots: {sum(1 for n in nodes if inn[n] == 0)}")
