#!/usr/bin/env python3
"""Which Python call sites the stock-torch device time of a plain training iteration comes from (GPU box only):

    python tools/torch_ops_by_site.py > gpurun_out/torch_ops_by_site.txt

torch.profiler with input shapes over three plain iterations; per (aten op, input shapes) the device time per iteration,
largest first."""
import collections
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

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
for _ in range(3):
    trainer.train_iteration(real)
torch.cuda.synchronize()
ITERS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(ITERS):
        trainer.train_iteration(real)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", 0.0) or getattr(ev, "self_cuda_time_total", 0.0)
    if t <= 0 or not ev.key.startswith("aten::"):
        continue
    site = str(ev.input_shapes)
    rows[(ev.key, site)][0] += t
    rows[(ev.key, site)][1] += ev.count
    total += t
print(f"stock aten ops with device time: {total / ITERS / 1e3:.2f} ms per iteration")
for (op, site), (t, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{t / ITERS / 1e3:7.3f} ms  {n / ITERS:6.1f} calls  {op:28s} {site[:150]}")
