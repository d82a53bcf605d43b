#!/usr/bin/env python3
"""Why the dominant conv launch runs 1 300 TFLOP/s inside the training step and 1 500 back to back: the same launch (3x3
512 -> 512 @256^2, per-sample weights, batch 16) timed (a) back to back, (b) alternating with an HBM-bound kernel (the 4x4
blur of a 512-channel 256^2 map), (c) alternating with the kernel that writes its weights (msg_modulate_weights), (d) with
fresh (never read) input and weights every launch -- HIP events around the conv launches only.  GPU box."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import conv_ops                                            # noqa: E402
from multi_stylegan_amd.op_static import upfirdn2d                                 # noqa: E402

DEV = "cuda:0"
b, i, o, r, k = 16, 512, 512, 256, 3
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
xs = [cl(torch.randn(b, i, r, r, device=DEV, dtype=torch.bfloat16)) for _ in range(3)]
w = torch.randn(b, o, i, k, k, device=DEV) / math.sqrt(i * k * k)
wks = [conv_ops._relay_fwd(w * (1 + 0.01 * j), torch.bfloat16) for j in range(3)]
fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64).to(DEV)
xb = cl(torch.randn(b, 512, 257, 257, device=DEV, dtype=torch.bfloat16))
flops = 2.0 * b * r * r * o * i * k * k


def conv(j=0):
    return conv_ops._launch_fprop(xs[j % 3], wks[j % 3][0], wks[j % 3][1], None, o, (r, r), k, k, 1, 1, 1, False, True, i)


def run(name, between, n=40, rotate=False):
    for j in range(5):
        conv(j if rotate else 0); between()
    evs = []
    for j in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); conv(j if rotate else 0); e.record()
        evs.append((a, e))
        between()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(e) for a, e in evs)
    t = ts[len(ts) // 2] * 1e-3
    print(f"{name:60s} {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s", flush=True)


with torch.no_grad():
    for rep in range(2):
        run("back to back, same operands", lambda: None)
        run("alternating with the 4x4 blur of a 512ch 257^2 map", lambda: upfirdn2d(xb, fir, pad=(1, 1)))
        run("alternating with 3 blurs (~1.4 ms of HBM-bound work)", lambda: [upfirdn2d(xb, fir, pad=(1, 1)) for _ in range(3)])
        run("rotating over 3 inputs / weight sets (3.4 GB: nothing warm)", lambda: None, rotate=True)
        run("rotating operands + a blur in between", lambda: upfirdn2d(xb, fir, pad=(1, 1)), rotate=True)

# (e) sustained: the same launch back to back for ~20 s, rate per block of 300 launches -- does the burst rate hold once the
# chip has been at full load for as long as a benchmark run lasts?
if os.environ.get("INSTEP_SUSTAINED", "1") != "0":
    with torch.no_grad():
        for block in range(20):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(300):
                conv(0)
            e.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(e) * 1e-3 / 300
            print(f"sustained block {block:2d} ({(block + 1) * 300 * t:5.1f} s in): {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s", flush=True)
