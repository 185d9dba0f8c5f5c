import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as entry
from oracle.golden_recipe import golden_inputs
sys.path.insert(0, "tests")
batch_mod = importlib.import_module("ts-asr_amd.batch")
inp = golden_inputs()
T = lambda k: torch.from_numpy(inp[k])
def make_batch():
    return batch_mod.PaddedBatch({
        "id": ["a", "b", "c", "d"],
        "mixed_sig": batch_mod.PaddedData(T("mixed_sig"), T("mixed_lens")),
        "enroll_sig": batch_mod.PaddedData(T("enroll_sig"), T("enroll_lens")),
        "tokens_bos": batch_mod.PaddedData(T("tokens_bos"), T("tokens_bos_lens")),
        "tokens": batch_mod.PaddedData(T("tokens"), T("tokens_lens")),
    })
accum = int(os.environ.get("ACCUM", "2"))
runs = {}
for trial in range(int(os.environ.get("TRIALS", "5"))):
    for mode in ("eager", "graph"):
        brain, h = entry._config1_brain("cuda:0", "bf16")
        brain.grad_accumulation_factor = accum
        brain.overlap_branches = os.environ.get("OVERLAP", "1") == "1"
        brain.modules.train()
        if mode == "graph":
            brain.enable_hip_graph(warmup_steps=2)
        batch = make_batch().to("cuda:0")
        ls = [float(brain.fit_batch(batch)) for _ in range(8)]
        runs.setdefault(mode, []).append(ls)
        print(trial, mode, ["%.6f" % v for v in ls], flush=True)
for mode, rr in runs.items():
    print(mode, "all identical:", all(r == rr[0] for r in rr))
print("graph == eager:", runs["graph"][0] == runs["eager"][0])
