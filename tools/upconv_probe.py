#!/usr/bin/env python3
"""Is the pixel-shuffling store pattern what the sub-pixel up-convolution pays for?  The same contraction (512 -> 2048 on a
128^2 map, per-sample weights, batch 16) with and without the pixel-shuffling epilogue.  GPU box only."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import conv_ops

DEV = "cuda:0"
b, i, o, r = 16, 512, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = conv_ops.to_compute_layout(torch.randn(b, i, r, r, device=DEV), torch.bfloat16)
w = torch.randn(b, 4 * o, i, 1, 1, device=DEV) / math.sqrt(i)
wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


flops = 2.0 * b * r * r * 4 * o * i
for name, ps in (("pixel-shuffled [B,2H,2W,C]", True), ("plain [B,H,W,4C]", False)):
    fn = lambda: conv_ops._launch_fprop(x, wk, ck, None, 4 * o, (r, r), 1, 1, 1, 0, 1, ps, True, i)
    t = timeit(fn)
    print(f"{name:32s} {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s", flush=True)
