#!/usr/bin/env python3
"""Diagnostic: where a K-step of conv_wgrad_row3s_kernel spends its cycles, from in-kernel s_memtime stamps.
Build the stamped library beside the real one:
    python -m multi_stylegan_amd.build --variant stamps --flags=-DMSG_WGRAD3_STAMPS
    cp multi_stylegan_amd/libmsg_hip.so multi_stylegan_amd/libmsg_hip_stamps.so; python -m multi_stylegan_amd.build --force
then run  MSG_LIB_VARIANT=stamps python tools/wgrad3_stamps.py  on the GPU box (never ship or benchmark that build)."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops
b, i, o, r, k = 16, 512, 512, 256, 3
x = torch.randn(b, i, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
gy = torch.randn(b, o, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
for _ in range(100):
    gw = conv_ops._launch_wgrad(gy, x, o, i, k, k, 1, 1, False, True, None)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 4 * 2 * 8, dtype=np.uint64)
assert h.msg_wgrad3_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
s = buf.reshape(256, 4, 2, 8).astype(np.int64)
names = ["sub-step 0, MFMAs 0..31 (32 fragment reads)", "sub-step 0, MFMAs 32..47", "lgkmcnt wait", "barrier",
         "sub-step 1, MFMAs 0..29 (30 reads)", "sub-step 1, MFMAs 30..38 (2 reads, 5 DMA pieces)", "sub-step 1, MFMAs 39..47 (4 DMA pieces)"]
d = np.diff(s, axis=3)
for st in range(2):
    print(f"K-step {8 + st}")
    for nm, v, q in zip(names, np.median(d[:, :, st, :].reshape(-1, 7), axis=0), np.percentile(d[:, :, st, :].reshape(-1, 7), 90, axis=0)):
        print(f"   {nm:48s} median {v:7.0f}   p90 {q:7.0f} cycles")
    print(f"   whole step: median {np.median(s[:, :, st, 7] - s[:, :, st, 0]):.0f} cycles (96 MFMAs = 1536 MFMA cycles; 16 cycles per MFMA + ~70 per stamp)")
