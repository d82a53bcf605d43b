#!/usr/bin/env python3
"""One styled 3x3 layer of the generator's last level (512 -> 512 @256^2, batch 16, bf16) through the product's own autograd
op, every launch timed with HIP events (conv_ops kernel clock): forward alone (the conv with the fused noise + bias + leaky
ReLU epilogue and the sign bytes), then forward + backward (data gradient, per-sample weight gradient, modulation backward).
Answers which of the step's five launches of the dominant kernel run below its back-to-back rate.  GPU box."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops                                       # noqa: E402

DEV = "cuda:0"
b, c, r = 16, 512, 256
torch.manual_seed(0)
x = conv_ops.to_compute_layout(torch.randn(b, c, r, r, device=DEV), torch.bfloat16).requires_grad_(True)
w = torch.randn(1, c, c, 3, 3, device=DEV, requires_grad=True)
style = (1 + 0.3 * torch.randn(b, c, device=DEV)).requires_grad_(True)
bias = torch.zeros(c, device=DEV, requires_grad=True)
noise = torch.randn(b, 1, r, r, device=DEV)
nw = torch.full((1,), 0.3, device=DEV, requires_grad=True)
gy = conv_ops.to_compute_layout(torch.randn(b, c, r, r, device=DEV), torch.bfloat16)


def layer():
    return conv_ops.modulated_conv2d_bias_act(x, w, style, True, bias, noise, nw, 0.2, math.sqrt(2))


def report(title):
    torch.cuda.synchronize()
    print(title)
    for key, v in sorted(_lib.kernel_clock.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
        rate = v["work"] / (v["total_ms"] * 1e-3)
        unit = f"{rate / 1e12:8.1f} TFLOP/s" if ("conv" in key or "wgrad" in key) else f"{rate / 1e9:8.1f} GB/s"
        print(f"   {key:60s} x{v['launches']:3d}  {v['avg_us']:9.1f} us  {unit}")


for _ in range(3):
    layer().backward(gy)
os.environ["MSG_CLOCK_SHAPES"] = "1"
_lib.kernel_clock.reset(enabled=True)
with torch.no_grad():
    for _ in range(20):
        layer()
report("forward only (x20): conv + fused noise / bias / leaky ReLU + sign bytes")
_lib.kernel_clock.reset(enabled=True)
for _ in range(20):
    x.grad = w.grad = style.grad = bias.grad = nw.grad = None
    layer().backward(gy)
report("forward + backward (x20)")
