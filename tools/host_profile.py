#!/usr/bin/env python3
"""cProfile of the host side of the training step (where the Python time between launches goes).

    python tools/host_profile.py [--iters 4] > gpurun_out/host_profile.txt
"""
import argparse
import cProfile
import os
import pstats
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--resolution", type=int, default=256)
ap.add_argument("--single-thread", action="store_true",
                help="run backward on the calling thread (torch.autograd.set_multithreading_enabled(False)) so that cProfile sees "
                     "the Python backward functions too -- the engine's own thread is invisible to it")
args = ap.parse_args()

import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(args.resolution))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
random.seed(1)
real = torch.rand(args.batch, 2, 3, args.resolution, args.resolution, device=dev)
for _ in range(3):
    trainer.train_iteration(real)
torch.cuda.synchronize()
# host-only time: how long the Python side takes to ENQUEUE an iteration (the GPU runs behind)
t0 = time.perf_counter()
for _ in range(args.iters):
    trainer.train_iteration(real)
t_enqueue = (time.perf_counter() - t0) / args.iters
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / args.iters
print(f"enqueue {t_enqueue * 1e3:.1f} ms/iter, wall {t_total * 1e3:.1f} ms/iter")
pr = cProfile.Profile()
import contextlib
with (torch.autograd.set_multithreading_enabled(False) if args.single_thread else contextlib.nullcontext()):
    trainer.train_iteration(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        trainer.train_iteration(real)
    print(f"(this mode, unprofiled: enqueue {(time.perf_counter() - t0) / args.iters * 1e3:.1f} ms/iter)")
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(args.iters):
        trainer.train_iteration(real)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(70)
st.sort_stats("cumulative").print_stats(60)
