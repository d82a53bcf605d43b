#!/usr/bin/env python3
"""Diagnostic: phase timing of the ping-pong conv kernel from in-kernel s_memtime stamps.
Build the stamped library first:  python -m multi_stylegan_amd.build --variant stamps --flags=-DMSG_PP_STAMPS
(never ship or benchmark that build), then run this on the GPU box."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops
b, i, o, r, k = 16, 512, 512, 256, 3
x = torch.randn(b, i, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(b, o, i, k, k, device="cuda") / math.sqrt(i * k * k)
wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
for _ in range(3):
    y = conv_ops._launch_fprop(x, wk, ck, None, o, (r, r), k, k, 1, 1, 1, False, True, i)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 8 * 10, dtype=np.uint64)
assert h.msg_pp_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
s = buf.reshape(256, 8, 10).astype(np.int64)
names = ["A:dma-issue", "A:reads+wait", "A:barrier", "B:mfma", "B:barrier", "C:reads+wait", "C:vmcnt", "C:barrier", "D:mfma(+vmcnt g0)"]
d = np.diff(s, axis=2)                                  # [block, wave, 9]
for g, sl in (("group0 (waves 0-3)", slice(0, 4)), ("group1 (waves 4-7)", slice(4, 8))):
    print(g)
    for n, v in zip(names, np.median(d[:, sl, :].reshape(-1, 9), axis=0)):
        print(f"   {n:22s} {v:8.0f} cycles")
print("one K-tile, wave 0 (A start -> D end): median", np.median(s[:, 0, 9] - s[:, 0, 0]))
print("offset group1 - group0 at phase A start:", np.median(s[:, 4, 0] - s[:, 0, 0]))
