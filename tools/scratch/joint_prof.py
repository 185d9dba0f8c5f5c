import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
rn = importlib.import_module("ts-asr_amd.rnnt")
DEV = "cuda"
B, T, U1, J, V = 32, 250, 121, 640, 29
g = torch.Generator().manual_seed(0)
enc = torch.randn(B, T, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
dec = torch.randn(B, U1, J, generator=g).to(DEV, torch.bfloat16).requires_grad_()
W = (torch.randn(V, J, generator=g) / J ** 0.5).to(DEV).requires_grad_()
b = torch.zeros(V, device=DEV, requires_grad=True)
tg = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).to(DEV)
tl = torch.full((B,), T, dtype=torch.int32, device=DEV); ul = torch.full((B,), U1 - 1, dtype=torch.int32, device=DEV)
for it in range(6):
    for t in (enc, dec, W, b): t.grad = None
    logits = rn.fused_joint_logits(enc, dec, W, b, 0.01, tl, ul)
    loss = rn.rnnt_costs(logits, tg, tl, ul, 0).mean()
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
