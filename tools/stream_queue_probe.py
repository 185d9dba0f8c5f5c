#!/usr/bin/env python3
"""Which torch streams run concurrently with which (streams share the runtime's hardware queues: GPU_MAX_HW_QUEUES, default 4)?
A spin kernel on stream i, a tiny kernel on stream j right after: j is on another hardware queue iff the tiny kernel finishes first."""
import os, sys
import torch
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.zeros(64, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
torch.cuda._sleep(1000); torch.cuda.synchronize()


def concurrent(i, j):
    a0, a1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[i]):
        a0.record()
        torch.cuda._sleep(2_000_000)
        a1.record()
    with torch.cuda.stream(streams[j]):
        x.add_(1.0)
        b1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(b1) < 0.5 * a0.elapsed_time(a1), a0.elapsed_time(a1)


print("spin kernel: %.2f ms" % concurrent(0, 1)[1])
for i in range(n):
    print("stream %2d concurrent with:" % i, " ".join(str(j) for j in range(n) if j != i and concurrent(i, j)[0]))
