#!/usr/bin/env python3
"""The modulated conv of the generator's low-resolution layers in its two forms: per-sample weights (MSG_MODCONV_SMALL_MAP=0)
and activation scaling (conv_ops._ModulatedConvSmall).  GPU box:  python tools/smallmap_probe.py"""
import math, sys, torch
sys.path.insert(0, '.')
from multi_stylegan_amd import conv_ops
DEV = 'cuda:0'
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 16
for hw in (4, 8, 16, 32):
    x = conv_ops.to_compute_layout(torch.randn(B, 512, hw, hw, device=DEV), torch.bfloat16).requires_grad_(True)
    w = torch.randn(1, 512, 512, 3, 3, device=DEV, requires_grad=True)
    st = (1 + 0.1 * torch.randn(B, 512, device=DEV)).requires_grad_(True)
    bias = torch.zeros(512, device=DEV, requires_grad=True)
    noise = torch.randn(B, 1, hw, hw, device=DEV)
    nw = torch.zeros(1, device=DEV, requires_grad=True)
    gy = conv_ops.to_compute_layout(torch.randn(B, 512, hw, hw, device=DEV), torch.bfloat16)
    for pixels in (0, 4096):
        conv_ops._SMALL_MAP_PIXELS = pixels
        fwd = lambda: conv_ops.modulated_conv2d_bias_act(x, w, st, True, bias, noise, nw, scale=math.sqrt(2))
        def both():
            y = fwd()
            torch.autograd.grad(y, (x, w, st, bias, nw), gy)
        with torch.no_grad():
            tf = timeit(fwd)
        print(f"{hw:3d}^2  {'scaling   ' if pixels else 'per-sample'}  fwd {tf:7.1f} us   fwd+bwd {timeit(both):7.1f} us", flush=True)
