#!/usr/bin/env python3
"""Does an HBM-bound kernel hide behind an MFMA-bound one when they run on two HIP streams?  (Round 5: ~15 ms of a 99 ms
iteration are HBM-bound launches -- activation backward, FIR passes, reductions -- that sit BETWEEN the contraction kernels of
one stream; the weight gradients of backward do not depend on the data-gradient chain and could run beside it.)

The dominant conv launch (3x3 512 -> 512 @256^2, per-sample weights, batch 16: one workgroup per CU, all registers) and the
4x4 blur of a 512-channel 256^2 map, n launches each:  (a) all on one stream, (b) convs on one stream and blurs on another,
(c) the same with the weight-gradient kernel in the conv's place.  Wall time per (conv + blur) pair.  GPU box."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import conv_ops                                            # noqa: E402
from multi_stylegan_amd.op_static import upfirdn2d                                 # noqa: E402

DEV = "cuda:0"
b, i, o, r, k = 16, 512, 512, 256, 3
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(b, i, r, r, device=DEV, dtype=torch.bfloat16))
gy = cl(torch.randn(b, o, r, r, device=DEV, dtype=torch.bfloat16))
w = torch.randn(b, o, i, k, k, device=DEV) / math.sqrt(i * k * k)
wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64).to(DEV)
xb = cl(torch.randn(b, 512, 257, 257, device=DEV, dtype=torch.bfloat16))
side = torch.cuda.Stream()


def conv():
    return conv_ops._launch_fprop(x, wk, ck, None, o, (r, r), k, k, 1, 1, 1, False, True, i)


def wgrad():
    return conv_ops._launch_wgrad(gy, x, o, i, k, k, 1, 1, False, True, None, raw=True)


def blur():
    return upfirdn2d(xb, fir, pad=(1, 1))


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def one_stream(mm, n, blurs_per):
    def go():
        for _ in range(n):
            mm()
            for _ in range(blurs_per):
                blur()
    return go


def two_streams(mm, n, blurs_per):
    def go():
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(n):
            mm()
        with torch.cuda.stream(side):
            for _ in range(n * blurs_per):
                blur()
        torch.cuda.current_stream().wait_stream(side)
    return go


n = 12
for name, mm in (("conv_fprop_row3<4,4> 512->512 @256^2", conv), ("conv_wgrad_row3s 512->512 @256^2", wgrad)):
    t_mm = timed(lambda: [mm() for _ in range(n)]) / n
    t_bl = timed(lambda: [blur() for _ in range(n)]) / n
    for per in (1, 4):
        t1 = timed(one_stream(mm, n, per)) / n
        t2 = timed(two_streams(mm, n, per)) / n
        print(f"{name}: alone {t_mm:.3f} ms, blur alone {t_bl:.3f} ms; + {per} blur(s): one stream {t1:.3f} ms per group, "
              f"two streams {t2:.3f} ms  (hidden: {100 * (t1 - t2) / (per * t_bl):.0f} % of the blur time)")
