#!/usr/bin/env python3
"""Diagnostic: where a K-step of conv_fprop_row3_kernel<4,4> spends its cycles, from in-kernel s_memtime stamps.
Build the stamped library first:  python -m multi_stylegan_amd.build --variant stamps --flags=-DMSG_ROW3_STAMPS
(never ship or benchmark that build), then run this on the GPU box."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_stylegan_amd import _lib, conv_ops
# ROW3_STAMPS_SHAPE="batch,in,out,resolution,per_sample" (default: the 512 -> 512 @256^2 per-sample launch of the 256 x 256 tile;
# "32,128,128,256,0" is the discriminator's 128 -> 128 @256^2 layer on the 128 x 128 tile: 512 MFMA cycles per K-step there)
b, i, o, r, ps = (int(v) for v in os.environ.get("ROW3_STAMPS_SHAPE", "16,512,512,256,1").split(","))
k = 3
mfma_cycles = 2048 if (o >= 256 and o % 256 == 0 and i > 128) else 512     # 256 x 256 tile: 128 MFMAs of 16 cycles per wave and K-step; 128 x 128: 32
x = torch.randn(b, i, r, r, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn((b, o, i, k, k) if ps else (o, i, k, k), device="cuda") / math.sqrt(i * k * k)
wk, ck = conv_ops._relay_fwd(w, torch.bfloat16)
for _ in range(200):          # long enough for the clock to settle
    y = conv_ops._launch_fprop(x, wk, ck, None, o, (r, r), k, k, 1, 1, 1, False, bool(ps), i)
torch.cuda.synchronize()
h = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(256 * 4 * 3 * 8, dtype=np.uint64)
assert h.msg_row3_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
s = buf.reshape(256, 4, 3, 8).astype(np.int64)
s16 = os.environ.get("MSG_CONV_ROW3_S16", "1") != "0"
# stamp slots: 0 step start | 1..3 after sub-steps 0..2 (not the last) | 4 after the wait | 5 after the barrier | 6 after the last sub-step
order = [0, 1, 4, 5, 6] if s16 else [0, 1, 2, 3, 4, 5, 6]
names = (["sub-step 0 (64 MFMAs)", "vmcnt/lgkmcnt wait", "barrier", "sub-step 1 (64 MFMAs) + next step's first reads"] if s16 else
         ["kk=0", "kk=1", "kk=2", "vmcnt/lgkmcnt wait", "barrier", "kk=3 + next step's first reads"])
d = np.diff(s[..., order], axis=3)
n = len(names)
for st, label in enumerate(("K-step 9  (kw=0: next activation tile issued)", "K-step 10 (kw=1)", "K-step 11 (kw=2)")):
    print(label)
    for nm, v, q in zip(names, np.median(d[:, :, st, :].reshape(-1, n), axis=0), np.percentile(d[:, :, st, :].reshape(-1, n), 90, axis=0)):
        print(f"   {nm:48s} median {v:7.0f}   p90 {q:7.0f} cycles")
    print(f"   whole step: median {np.median(s[:, :, st, 6] - s[:, :, st, 0]):.0f} cycles ({mfma_cycles} MFMA cycles per wave"
          + ("" if o >= 256 else "; two workgroups share the SIMDs: 1024 per K-step pair") + ")")
    for w in range(4):
        print(f"      wave {w}: " + " ".join(f"{v:6.0f}" for v in np.median(d[:, w, st, :], axis=0)))
print("three steps, start to start:", np.median(s[:, :, 2, 0] - s[:, :, 0, 0]) / 2)

# in-kernel clock over the whole K loop: shader cycles per 100 MHz reference tick (MI355X_MICROARCH.md, DVFS give-back item 6)
cbuf = np.zeros(256 * 4 * 8, dtype=np.uint64)
if hasattr(h, "msg_row3_clock_read") and h.msg_row3_clock_read(cbuf.ctypes.data_as(ctypes.c_void_p), cbuf.nbytes) == 0:
    # stamps of the LAST 256 workgroups of the launch: 0 kernel entry | 1 K loop start | 2 K loop end | 3 kernel exit
    c = cbuf.reshape(256, 4, 4, 2).astype(np.int64)
    ok = (c[:, :, 3, 1] > c[:, :, 0, 1]) & (c[:, :, 0, 1] > 0)
    for nm, a, z in (("entry -> K loop (offsets, noise/bias staging, first loads issued)", 0, 1), ("K loop", 1, 2),
                     ("K loop end -> exit (epilogue)", 2, 3), ("whole workgroup", 0, 3)):
        cyc, tick = (c[:, :, z, 0] - c[:, :, a, 0])[ok], (c[:, :, z, 1] - c[:, :, a, 1])[ok]
        print(f"   {nm:68s} median {np.median(cyc):8.0f} cycles = {np.median(tick) * 1e-2:6.1f} us")
    cycles, ticks = (c[:, :, 2, 0] - c[:, :, 1, 0])[ok], (c[:, :, 2, 1] - c[:, :, 1, 1])[ok]
    ghz = cycles / ticks * 0.1
    print(f"in-kernel clock over the K loop: median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}); "
          f"K loop median {np.median(cycles):.0f} cycles = {np.median(ticks) * 1e-2:.1f} us")
