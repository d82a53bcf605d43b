#!/usr/bin/env python3
"""The row-sharing 3x3 kernels (conv_fprop_row3_kernel<4,4> / <2,2>) on the shapes that carry their time, back to back, HIP
events around every launch: median / min time and TFLOP/s per shape.  For the kernel-row-order experiment (round 5):

    MSG_LIB_VARIANT=tuning MSG_ROW3_KH_ORDER=0 python tools/row3_order_probe.py     # every tile kh = 0, 1, 2
    MSG_LIB_VARIANT=tuning MSG_ROW3_KH_ORDER=1 python tools/row3_order_probe.py     # per-tile order (the product)

and, for the bytes each order moves beyond L2 (separate processes, counters only):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -- python tools/row3_order_probe.py --launches 6
    python tools/pmc_kernel.py <dir> conv_fprop_row3
GPU box."""
import argparse
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--only", default="")
args = ap.parse_args()
DEV = "cuda:0"
#        name                        B   I    O    R   per-sample
SHAPES = [("512->512 @256 ps B16", 16, 512, 512, 256, True),
          ("512->512 @128 ps B16", 16, 512, 512, 128, True),
          ("512->512 @64 ps B16", 16, 512, 512, 64, True),
          ("256->256 @128 B32", 32, 256, 256, 128, False),
          ("128->128 @256 B32", 32, 128, 128, 256, False),
          ("128->256 @256 B32", 32, 128, 256, 256, False),
          ("256->128 @256 B32", 32, 256, 128, 256, False),
          ("256->384 @128 B32", 32, 256, 384, 128, False),
          ("768->768 @32 B32", 32, 768, 768, 32, False)]
print("library:", os.path.basename(_lib.LIB_PATH), " MSG_ROW3_KH_ORDER =", os.environ.get("MSG_ROW3_KH_ORDER", "(default)"))
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, b, i, o, r, ps in SHAPES:
    if args.only and args.only not in name:
        continue
    torch.manual_seed(0)
    x = cl(torch.randn(b, i, r, r, device=DEV, dtype=torch.bfloat16))
    w = torch.randn((b, o, i, 3, 3) if ps else (o, i, 3, 3), device=DEV) / math.sqrt(9 * i)
    wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
    fn = lambda: conv_ops._launch_fprop(x, wk, ck, None, o, (r, r), 3, 3, 1, 1, 1, False, ps, i)
    plan = _lib.lib().msg_conv2d_fprop_plan(_lib.MSG_BF16, b, r, r, i, ck, r, r, o, 3, 3, wk.stride(0) if ps else 0)
    for _ in range(3):
        y = fn()
    evs = []
    for _ in range(args.launches):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); e.record()
        evs.append((a, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(e) * 1e3 for a, e in evs)
    fl = 2.0 * b * r * r * o * i * 9
    med = statistics.median(ts)
    print(f"{name:24s} plan {plan}  median {med:8.1f} us  min {ts[0]:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s (median)  "
          f"checksum {float(y.float().abs().mean()):.6f}")
    del x, w, wk, y
    torch.cuda.empty_cache()
