#!/usr/bin/env python3
"""msg_bias_act_backward_mask_head (the activation backward that forms the image head's data gradient itself) at the
generator's shapes, against the plain masked activation backward on the same map: us and GB/s.  GPU box."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd.op_static import fused_act                                   # noqa: E402

DEV = "cuda:0"
bf = torch.bfloat16


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for b, c, hw, other in ((16, 512, 256, False), (16, 512, 128, True), (16, 512, 64, True), (16, 512, 32, True), (16, 512, 16, True)):
    y = torch.randn(b, c, hw, hw, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
    mbytes = torch.randint(0, 256, (b * hw * hw, c // 8), device=DEV, dtype=torch.uint8)
    gy = torch.randn(b, c, hw, hw, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
    hbuf = torch.randn(b, hw, hw, 8, device=DEV).to(bf)
    ghead = hbuf.permute(0, 3, 1, 2)[:, :6]
    whead, style = torch.randn(6, c, device=DEV), 1 + 0.3 * torch.randn(b, c, device=DEV)
    noise = torch.randn(b, 1, hw, hw, device=DEV)
    for tile in ((1, c), (256, 256)):
        if (b * hw * hw) % tile[0]:
            continue
        t = timed(lambda: fused_act.act_backward_with_head(gy if other else None, (ghead, whead, style, 0.044), (b, c, hw, hw),
                                                           noise, None, True, 0.2, math.sqrt(2), (mbytes, *tile)))
        nbytes = (1 + other) * y.numel() * 2 + mbytes.numel() + b * hw * hw * 16
        t2 = timed(lambda: fused_act.FusedLeakyReLUFunctionBackward.apply(gy, y, noise, True, 0.2, math.sqrt(2), (mbytes, *tile)))
        print(f"B{b} {c}ch {hw}^2 other={other} mask tiles {tile}: head-fused {t:7.1f} us ({nbytes / t / 1e3:6.0f} GB/s)   "
              f"plain masked backward {t2:7.1f} us ({(2 * y.numel() * 2 + mbytes.numel()) / t2 / 1e3:6.0f} GB/s)")
