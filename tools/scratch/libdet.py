"""Is the library fp32 GEMM of the LSTM input side bit-reproducible run to run? (dw_ih = dg^T . x, K = B*U)"""
import torch
torch.manual_seed(0)
for (BU, H4, I) in [(32 * 121, 2048, 28), (4 * 13, 4 * 64, 28), (4 * 13, 4 * 32, 28), (2 * 9, 256, 28), (32 * 121, 2048, 32)]:
    dg = torch.randn(BU, H4, device="cuda"); x = torch.randn(BU, I, device="cuda"); w = torch.randn(H4, I, device="cuda"); b = torch.randn(H4, device="cuda")
    ref1, ref2, ref3 = dg.t() @ x, dg @ w, torch.nn.functional.linear(x, w, b)
    bad = [0, 0, 0]
    for _ in range(30):
        bad[0] += int(not torch.equal(dg.t() @ x, ref1)); bad[1] += int(not torch.equal(dg @ w, ref2))
        bad[2] += int(not torch.equal(torch.nn.functional.linear(x, w, b), ref3))
    print((BU, H4, I), "non-identical repeats (dw, dx, fwd):", bad, flush=True)
