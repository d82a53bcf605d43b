#!/usr/bin/env python3
"""The separable 4x4 blur (csrc/blur_sep.hip) at the shapes of the training step, plain and with the fused noise + bias +
leaky ReLU stage, forward only: median time and algorithmic GB/s.  GPU box:  [MSG_LIB_VARIANT=<tag>] python tools/blur_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit, cl                                           # noqa: E402
from multi_stylegan_amd.op_static import blur_bias_act, upfirdn2d                  # noqa: E402

DEV = "cuda:0"
fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64 * 4).to(DEV)
B = int(os.environ.get("BLUR_PROBE_BATCH", "16"))
print("library:", os.environ.get("MSG_LIB_VARIANT", "(product)"))
# G: blur behind the up-conv, [B, 512, 2R+1, 2R+1] -> [B, 512, 2R, 2R] pad (1, 1), with the activation; D: blur behind the
# strided conv [B, C, R-1, R-1] -> [B, C, R, R] pad (2, 2), plain; backward of both = plain blur of the gradient
cases = [("G 512ch ->256^2 +act", (B, 512, 257, 257), (1, 1), True), ("G 512ch ->128^2 +act", (B, 512, 129, 129), (1, 1), True),
         ("G 512ch ->64^2 +act", (B, 512, 65, 65), (1, 1), True), ("G 512ch ->32^2 +act", (B, 512, 33, 33), (1, 1), True),
         ("G grad 512ch 256^2", (B, 512, 256, 256), (2, 2), False), ("G grad 512ch 128^2", (B, 512, 128, 128), (2, 2), False),
         ("D 128ch 127->128", (2 * B, 128, 127, 127), (2, 2), False), ("D 256ch 63->64", (2 * B, 256, 63, 63), (2, 2), False),
         ("D 384ch 31->32", (2 * B, 384, 31, 31), (2, 2), False)]
with torch.no_grad():
    for name, shape, pad, act in cases:
        x = cl(torch.randn(*shape, device=DEV, dtype=torch.bfloat16))
        if act:
            oh = shape[2] + pad[0] + pad[1] - 3
            bias, nz, nw = torch.randn(shape[1], device=DEV), torch.randn(shape[0], 1, oh, oh, device=DEV), torch.randn(1, device=DEV)
            fn = lambda: blur_bias_act(x, fir, pad, bias, nz, nw, 0.2, 2 ** 0.5)
        else:
            fn = lambda: upfirdn2d(x, fir, pad=pad)
        y = fn()
        nbytes = (x.numel() + y.numel()) * 2
        t = timeit(fn, 60, warm=5)
        print(f"{name:24s} {t * 1e6:8.1f} us  {nbytes / t / 1e9:7.1f} GB/s  ({nbytes / t / 8e12 * 100:4.1f} % of 8 TB/s)", flush=True)
        del x, y
