#!/usr/bin/env python3
"""Does a host->device copy on a side stream run BESIDE compute kernels on this box, or between them?  (GPU box only.)
A train of long-running kernels on the compute stream, with and without a concurrent 25 MB pinned H2D copy per kernel."""
import os
import sys
import time

import torch

dev = torch.device("cuda", 0)
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
host = torch.rand(16, 2, 3, 256, 256).pin_memory()
dst = torch.empty_like(host, device=dev)
side = torch.cuda.Stream()


def run(copies, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        torch.mm(a, b)
        if copies:
            with torch.cuda.stream(side):
                dst.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(False, 5)
for _ in range(3):
    print(f"mm only {run(False):.3f} ms   mm + concurrent 25 MB H2D {run(True):.3f} ms")
t0 = time.perf_counter()
for _ in range(20):
    dst.copy_(host, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D alone: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per 25 MB = {25.2 / ((time.perf_counter() - t0) / 20 * 1e3):.1f} GB/s")
