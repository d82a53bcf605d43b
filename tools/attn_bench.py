#!/usr/bin/env python3
"""Fused non-local attention (csrc/attention.hip) against the composite path (library bmm + softmax kernel + bmm):
forward and forward+backward time, TFLOP/s on the algorithmic FLOPs 2 B Nq Nk (dk + dv) [fwd] / x3.5 [fwd+bwd].
GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd.op_static import attention, softmax_rows  # noqa: E402


def composite(q, k, v):
    return torch.bmm(softmax_rows(torch.bmm(q, k.transpose(1, 2))), v)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for (b, nq, nk) in ((16, 4096, 1024), (32, 4096, 1024), (8, 16384, 4096)):
    if os.environ.get("ATTN_BENCH_BF16_ONLY"):
        pass
    for dtype in ((torch.bfloat16,) if os.environ.get("ATTN_BENCH_BF16_ONLY") else (torch.bfloat16, torch.float32)):
        q = (torch.randn(b, nq, 48, device="cuda") * 0.5).to(dtype).requires_grad_(True)
        k = (torch.randn(b, nk, 48, device="cuda") * 0.5).to(dtype).requires_grad_(True)
        v = torch.randn(b, nk, 192, device="cuda").to(dtype).requires_grad_(True)
        go = torch.randn(b, nq, 192, device="cuda").to(dtype)
        flops = 2.0 * b * nq * nk * (48 + 192)
        for name, fn in (("fused", attention.non_local_attention), ("composite", composite)):
            with torch.no_grad():
                tf = timeit(lambda: fn(q, k, v))

            def both():
                o = fn(q, k, v)
                o.backward(go)
                q.grad = k.grad = v.grad = None
            tb = timeit(both)
            mem = torch.cuda.max_memory_allocated() / 2 ** 30
            print(f"B={b} {nq}x{nk} {str(dtype)[6:]:9s} {name:9s} fwd {tf:7.3f} ms ({flops / tf / 1e9:7.1f} TFLOP/s)  "
                  f"fwd+bwd {tb:7.3f} ms ({3.5 * flops / tb / 1e9:7.1f} TFLOP/s)  peak {mem:.1f} GiB", flush=True)
            torch.cuda.reset_peak_memory_stats()
