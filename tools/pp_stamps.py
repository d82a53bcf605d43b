#!/usr/bin/env python3
"""Diagnostic: phase timing of the ping-pong conv kernel from in-kernel s_memtime stamps.
Build the stamped library first:  python -m multi_stylegan_amd.build --variant stamps --flags=-DMSG_PP_STAMPS
(never ship or benchmark that build), then run this on the GPU box."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops
# a launch that the dispatcher gives to conv_fprop_pp_kernel: the data gradient of the 2x up-convolution (2x2, stride 2,
# 512 -> 512, 256^2 -> 128^2, per-sample weights; 32 K-tiles)
b, i, o, r, k, st = 16, 512, 512, 256, 2, 2
x = torch.randn(b, i, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(b, o, i, k, k, device="cuda") / math.sqrt(i * k * k)
wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
for _ in range(int(os.environ.get("PP_STAMPS_LAUNCHES", "40"))):
    y = conv_ops._launch_fprop(x, wk, ck, None, o, (r // st, r // st), k, k, st, 0, 1, False, True, i)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 8 * 10, dtype=np.uint64)
assert h.msg_pp_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
s = buf.reshape(256, 8, 10).astype(np.int64)
s = s[s[:, 0, 0] > 0]                                   # (a launch with fewer than 256 workgroups per sample fills only the last ones)
names = ["A:dma-issue", "A:reads+wait", "A:barrier", "B:mfma", "B:barrier", "C:reads+wait", "C:vmcnt", "C:barrier", "D:mfma(+vmcnt g0)"]
d = np.diff(s, axis=2)                                  # [block, wave, 9]
print(f"K-tile 8 of the last {len(s)} workgroups of the launch (cycles, median)")
for g, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
    print(g)
    for n, v in zip(names, np.median(d[:, sl, :].reshape(-1, 9), axis=0)):
        print(f"   {n:22s} {v:8.0f} cycles")
print("one K-tile, wave 0 (A start -> D end): median", np.median(s[:, 0, 9] - s[:, 0, 0]))
print("offset group1 - group0 at phase A start:", np.median(s[:, 4, 0] - s[:, 0, 0]))

cbuf = np.zeros(256 * 8 * 8, dtype=np.uint64)
if hasattr(h, "msg_pp_clock_read") and h.msg_pp_clock_read(cbuf.ctypes.data_as(ctypes.c_void_p), cbuf.nbytes) == 0:
    # the LAST 256 workgroups of the launch: 0 kernel entry | 1 K loop start | 2 K loop end | 3 kernel exit
    c = cbuf.reshape(256, 8, 4, 2).astype(np.int64)
    ok = (c[:, :, 3, 1] > c[:, :, 0, 1]) & (c[:, :, 0, 1] > 0)
    for nm, a, z in (("entry -> K loop (coordinates, first K-tile requested and landed)", 0, 1), ("K loop", 1, 2),
                     ("K loop end -> exit (epilogue)", 2, 3), ("whole workgroup", 0, 3)):
        cyc, tick = (c[:, :, z, 0] - c[:, :, a, 0])[ok], (c[:, :, z, 1] - c[:, :, a, 1])[ok]
        print(f"   {nm:68s} median {np.median(cyc):8.0f} cycles = {np.median(tick) * 1e-2:6.1f} us")
    cycles, ticks = (c[:, :, 3, 0] - c[:, :, 0, 0])[ok], (c[:, :, 3, 1] - c[:, :, 0, 1])[ok]
    print(f"in-kernel clock: median {np.median(cycles / ticks * 0.1):.3f} GHz")

pbuf = np.zeros(256 * 8 * 16, dtype=np.uint64)
if hasattr(h, "msg_pp_period_read") and h.msg_pp_period_read(pbuf.ctypes.data_as(ctypes.c_void_p), pbuf.nbytes) == 0:
    q = pbuf.reshape(256, 8, 16).astype(np.int64)
    q = q[q[:, 0, 0] > 0]
    per = np.diff(q, axis=2)
    print("K-tile period (start to start), K-tiles 8..22, median over waves:", " ".join(f"{v:.0f}" for v in np.median(per.reshape(-1, 15), axis=0)))
if os.environ.get("PP_STAMPS_RAW"):
    print("raw, last workgroup, wave 0: clock stamps (entry, loop start, loop end, exit):", c[255, 0, :, 0] - c[255, 0, 0, 0])
    print("   K-tile starts 8..23 relative to loop start:", q[-1, 0, :] - c[255, 0, 1, 0])
    print("   phase stamps of K-tile 8 relative to loop start:", s[-1, 0, :] - c[255, 0, 1, 0])
