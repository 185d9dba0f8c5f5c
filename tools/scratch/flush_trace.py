"""Who flushes the deferred-reduction queue mid-step? (eager step of the bench config; prints each flush with pending jobs, the stream and the caller)"""
import importlib, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bench = importlib.import_module("bench")
ops = importlib.import_module(bench.PKG + ".ops"); C = importlib.import_module(bench.PKG + "._capi")
batch_mod = importlib.import_module(bench.PKG + ".batch")
brain, h, _ = bench.build_brain("cuda:0", "bf16", 1)
batch = batch_mod.synthetic_batch(bench.B_LOCAL, bench.T_MEL, bench.T_ENROLL, bench.U, feats=True, seed=1234).to("cuda:0")
for _ in range(2):
    brain.fit_batch(batch)
torch.cuda.synchronize()
orig = ops.reduce_flush
def traced():
    n = C.lib().tsasr_reduce_pending() if ops._DEFER["on"] else -1
    if n > 0:
        st = torch.cuda.current_stream()
        fr = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[-6:-1]]
        print("flush pending", n, "stream", hex(st.cuda_stream), " <- ".join(reversed(fr)), flush=True)
    return orig()
ops.reduce_flush = traced
print("main stream", hex(torch.cuda.current_stream().cuda_stream))
brain.fit_batch(batch)
torch.cuda.synchronize()
