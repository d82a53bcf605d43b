#!/usr/bin/env python3
"""Which call sites of this package copy big tensors in a training iteration (GPU box only): aten copy_/clone/_to_copy/cat
seen by a TorchDispatchMode (forward, the main thread) and by the same mode entered inside every custom backward
(autograd runs those on its own thread), attributed to the innermost frames inside the package.

    python tools/big_copies.py > gpurun_out/big_copies.txt
"""
import collections
import os
import random
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

SEEN = collections.defaultdict(lambda: [0, 0])
WATCH = ("copy_", "clone", "_to_copy", "cat", "contiguous", "add", "add_", "mul", "mul_", "sum", "fill_", "zero_", "zeros",
         "zeros_like", "sub", "div", "rsqrt", "pow", "neg", "where", "flip", "permute_copy", "bmm", "mm", "addmm")
REGULARISED = bool(int(os.environ.get("MSG_BIG_COPIES_REG", "0")))    # 1: survey a regularised iteration (R1 + path length)
MIN_ELEMS = int(os.environ.get("MSG_BIG_COPIES_MIN", str(1 << 20)))   # 0: every call, ranked by COUNT (the tiny-launch survey)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda and t.numel() >= MIN_ELEMS:
                frames = [f for f in traceback.extract_stack() if "multi_stylegan_amd" in f.filename and "tools" not in f.filename]
                site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1]) or "(outside the package)"
                key = (name, tuple(t.shape), str(t.dtype)[6:], site)
                SEEN[key][0] += 1
                SEEN[key][1] += t.numel() * t.element_size()
        return out


dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
random.seed(1)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
trainer.iteration = 16
for _ in range(2):
    trainer.train_iteration(real)
torch.cuda.synchronize()

# custom backward functions run on autograd's device thread: enter the mode there too
from torch.autograd.function import Function, BackwardCFunction
orig_apply = BackwardCFunction.apply


def spy_apply(self, *a):
    with Spy():
        return orig_apply(self, *a)


BackwardCFunction.apply = spy_apply
if REGULARISED:
    trainer.iteration = 15
with Spy():
    trainer.train_iteration(real)
torch.cuda.synchronize()
tot = sum(v[1] for v in SEEN.values())
print(f"{len(SEEN)} sites, {tot / 1e9:.2f} GB of outputs >= 1 Mi elements")
for (name, shape, dt, site), (n, b) in sorted(SEEN.items(), key=lambda kv: -(kv[1][0] if MIN_ELEMS == 0 else kv[1][1]))[:60]:
    print(f"{b / 1e6:9.1f} MB {n:4d}x {name:10s} {str(shape):26s} {dt:9s} {site}")
