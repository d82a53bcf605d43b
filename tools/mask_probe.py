#!/usr/bin/env python3
"""Cost of the sign bytes (ActEpilogue::mask) in the forward kernels that write them: the same launch with and without.
GPU box:  python tools/mask_probe.py"""
import math, sys, torch
sys.path.insert(0, '.')
from multi_stylegan_amd import conv_ops, _lib
from multi_stylegan_amd.op_static import fused_act, blur_bias_act
DEV = 'cuda:0'
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
bf = torch.bfloat16
for name, mk in [
    ("modconv 512->512 @256 B16", lambda: (conv_ops.to_compute_layout(torch.randn(16, 512, 256, 256, device=DEV), bf).requires_grad_(True),)),
    ("conv 128->128 @256 B32", lambda: (conv_ops.to_compute_layout(torch.randn(32, 128, 256, 256, device=DEV), bf).requires_grad_(True),)),
    ("blur+act 512 @256 B16", lambda: (conv_ops.to_compute_layout(torch.randn(16, 512, 257, 257, device=DEV), bf).requires_grad_(True),)),
]:
    (x,) = mk()
    if name.startswith("modconv"):
        w = torch.randn(1, 512, 512, 3, 3, device=DEV, requires_grad=True); st = torch.ones(16, 512, device=DEV, requires_grad=True)
        bias = torch.zeros(512, device=DEV, requires_grad=True)
        f = lambda: conv_ops.modulated_conv2d_bias_act(x, w, st, True, bias, None, None, scale=math.sqrt(2))
    elif name.startswith("conv"):
        w = torch.randn(128, 128, 3, 3, device=DEV, requires_grad=True); bias = torch.zeros(128, device=DEV, requires_grad=True)
        f = lambda: conv_ops.conv2d_bias_act(x, w, bias, padding=1, scale=math.sqrt(2))
    else:
        fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 16).to(DEV)
        bias = torch.zeros(512, device=DEV, requires_grad=True)
        f = lambda: blur_bias_act(x, fir, (1, 1), bias, None, None, scale=math.sqrt(2))
    for flag in (False, True, False, True):
        fused_act.ACT_MASK = flag
        print(f"{name:28s} mask={int(flag)}: {timeit(f):9.1f} us", flush=True)
    del x
