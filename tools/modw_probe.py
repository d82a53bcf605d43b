#!/usr/bin/env python3
"""msg_modulate_weights (demodulation + per-sample weight set, one launch) at the generator's layer shapes: time and bytes/s.
GPU box:  [MSG_LIB_VARIANT=tuning MSG_MODW_MIN_WGS=...] python tools/modw_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit                                               # noqa: E402
from multi_stylegan_amd import _lib                                               # noqa: E402

DEV = "cuda:0"
print("library:", os.environ.get("MSG_LIB_VARIANT", "(product)"), "MSG_MODW_MIN_WGS =", os.environ.get("MSG_MODW_MIN_WGS", "(default)"))
# (batch, rows R, output channels O, input channels, taps): R = O for the plain convs, R = 4 O with one tap for the 2x2 up-conv
for b, r, o, i, t in ((16, 512, 512, 512, 9), (8, 512, 512, 512, 9), (16, 2048, 512, 512, 1), (16, 6, 6, 512, 1), (32, 512, 512, 512, 9)):
    base = torch.randn(r, t, i, device=DEV)
    wsq = base[:o].square().sum(dim=1).contiguous()
    style = 1 + 0.3 * torch.randn(b, i, device=DEV)
    out = torch.empty(b, r, t, i, device=DEV, dtype=torch.bfloat16)
    d = torch.empty(b, o, device=DEV)
    st = _lib.stream_of(torch.device(DEV))

    def fn():
        code = _lib.lib().msg_modulate_weights(base.data_ptr(), wsq.data_ptr(), style.data_ptr(), out.data_ptr(), d.data_ptr(),
                                               _lib.MSG_BF16, b, r, o, t, i, i, 0.05, 1e-8, st)
        _lib.check(code, "msg_modulate_weights")
    tt = timeit(fn, 200, warm=20)
    nbytes = out.numel() * 2 + base.numel() * 4
    print(f"B{b:3d} R{r:5d} O{o:4d} I{i} T{t}: {tt * 1e6:7.1f} us  {nbytes / tt / 1e9:7.1f} GB/s ({out.numel() * 2 / 1e6:.1f} MB written)", flush=True)
