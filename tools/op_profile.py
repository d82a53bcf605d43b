#!/usr/bin/env python3
"""torch.profiler view of the training step (which torch ops surround the HIP kernels, and what they cost).

    python tools/op_profile.py [--iters 4] > gpurun_out/op_profile.txt
"""
import argparse
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--resolution", type=int, default=256)
ap.add_argument("--stack", action="store_true")
ap.add_argument("--reg-only", action="store_true", help="profile exactly one iteration: the 16th (R1 + path length)")
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
for _ in range(15 if args.reg_only else 3):
    trainer.train_iteration(real)
torch.cuda.synchronize()
if args.reg_only:
    args.iters = 1
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=args.stack) as prof:
    for _ in range(args.iters):
        trainer.train_iteration(real)
    torch.cuda.synchronize()
print(f"# {args.iters} iterations (none of them a regularisation iteration unless iters >= 13)")
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70,
                                                         max_name_column_width=60, max_shapes_column_width=70))
if args.stack:
    # torch-op kernels only (ours are launched through ctypes and carry no aten op): name, shapes, python stack
    evs = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
           if e.key.startswith("aten::") and e.self_device_time_total > 0]
    evs.sort(key=lambda e: -e.self_device_time_total)
    counts = sorted(evs, key=lambda e: -e.count)
    print("# most frequent aten ops with device time (launch count matters for the host path)")
    for e in counts[:40]:
        print(f"  x{e.count / args.iters:6.1f}/iter {e.self_device_time_total / 1e3 / args.iters:7.3f} ms/iter  {e.key}  {str(e.input_shapes)[:90]}")
    for e in evs[:45]:
        print(f"{e.self_device_time_total / 1e3 / args.iters:8.3f} ms/iter  x{e.count / args.iters:6.1f}  {e.key}  {str(e.input_shapes)[:110]}")
        for fr in e.stack[:8]:
            if "multi_stylegan_amd" in fr or "bench" in fr or "tools" in fr:
                print("            ", fr[-110:])
