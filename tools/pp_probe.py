#!/usr/bin/env python3
"""Times the launches of the training step that run on conv_fprop_pp_kernel (256 x 256 tile, two wave groups in ping-pong) or,
`python tools/pp_probe.py generic`, on the LDS-DMA instantiation of the 128 x 128 kernel (conv_fprop.hip).
GPU box:  python tools/pp_probe.py [pp|generic]   (MSG_LIB_VARIANT to compare builds)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import conv_ops, _lib
DEV = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


# (name, batch, in, out columns, input map, kh, stride, pad, pixel_shuffle, per_sample)
cases = [("1x1 ps 512->2048 @64^2 per-sample", 16, 512, 2048, 64, 1, 1, 0, True, True),
         ("1x1 ps 512->2048 @32^2 per-sample", 16, 512, 2048, 32, 1, 1, 0, True, True),
         ("2x2 s2 512->512 256->128 per-sample", 16, 512, 512, 256, 2, 2, 0, False, True),
         ("3x3 385->768 @32^2 shared", 32, 385, 768, 32, 3, 1, 1, False, False),
         ("1x1 384->256 @128^2 shared", 32, 384, 256, 128, 1, 1, 0, False, False),
         ("1x1 384->768 @64^2 shared", 32, 384, 768, 64, 1, 1, 0, False, False),
         ("3x3 s2 256->256 128->63 shared", 32, 256, 256, 128, 3, 2, 0, False, False),
         # data gradients of the stride-2 convolutions in their parity form (2x2 taps, 4 I pixel-shuffled columns): 8 / 16 K-tiles
         ("2x2 ps 128->512 @127^2 shared", 32, 128, 512, 127, 2, 1, 1, True, False),
         ("2x2 ps 256->1024 @63^2 shared", 32, 256, 1024, 63, 2, 1, 1, True, False)]
generic = [("3x3 1024->1024 @16^2 shared", 32, 1024, 1024, 16, 3, 1, 1, False, False),
           ("3x3 768->385 @32^2 shared", 32, 768, 385, 32, 3, 1, 1, False, False),
           ("3x3 s2 128->128 256->127 shared", 32, 128, 128, 256, 3, 2, 0, False, False),
           ("3x3 512->512 @16^2 per-sample", 16, 512, 512, 16, 3, 1, 1, False, True),
           ("3x3 512->512 @8^2 per-sample", 16, 512, 512, 8, 3, 1, 1, False, True),
           ("3x3 512->512 @4^2 per-sample", 16, 512, 512, 4, 3, 1, 1, False, True),
           ("3x3 s2 384->384 64->31 shared", 32, 384, 384, 64, 3, 2, 0, False, False)]
short_k = [("1x1 128->256 @256^2 shared", 32, 128, 256, 256, 1, 1, 0, False, False),
           ("1x1 256->128 @256^2 shared", 32, 256, 128, 256, 1, 1, 0, False, False),
           ("1x1 128->256 @128^2 shared", 32, 128, 256, 128, 1, 1, 0, False, False),
           ("1x1 256->384 @128^2 shared", 32, 256, 384, 128, 1, 1, 0, False, False),
           ("1x1 512->128 @64^2 shared", 32, 512, 128, 64, 1, 1, 0, False, False),
           ("3x3 64->128 @64^2 shared", 32, 64, 128, 64, 3, 1, 1, False, False)]
if len(sys.argv) > 1 and sys.argv[1] == "generic":
    cases = generic
if len(sys.argv) > 1 and sys.argv[1] == "short":
    cases = short_k          # (register-staged by default: MSG_CONV_VARIANT=1 in a tuning build forces the LDS-DMA instantiation)
for name, b, i, n, r, k, s, pad, shuf, ps in cases:
    x = torch.randn(b, i, r, r, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn((b, n, i, k, k) if ps else (n, i, k, k), device=DEV) / math.sqrt(i * k * k)
    wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
    oh = (r + 2 * pad - k) // s + 1
    fn = lambda: conv_ops._launch_fprop(x, wk, ck, None, n, (oh, oh), k, k, s, pad, 1, shuf, ps, i)
    y = fn()
    t = timeit(fn)
    flops = 2.0 * b * oh * oh * n * i * k * k
    print(f"{name:40s} {t * 1e6:9.1f} us {flops / t / 1e12:8.1f} TFLOP/s", flush=True)
    del x, w, wk, y
