#!/usr/bin/env python3
"""Device time of stock torch / library kernels vs this package's HIP kernels, separately for plain iterations and for
the regularised (16th) iteration, with the biggest stock kernels listed.  GPU box only.

    python tools/torch_share.py > gpurun_out/torch_share.txt
"""
import collections
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

# stock = what tools/summarize_profiles.py counts as stock (torch's own kernels, library GEMMs, rocprim, runtime copies / fills);
# everything else is one of this package's kernels -- a list of OUR names went stale with every new kernel file
STOCK = ("at::native", "at_cuda_detail", "Cijk_", "rocprim", "Memcpy", "Memset", "__amd_rocclr", "hipcub", "elementwise_kernel")

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
random.seed(1)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
trainer.iteration = 13
for _ in range(4):                     # iterations 14, 15, 16 (regularised), 17: everything has run once
    trainer.train_iteration(real)
torch.cuda.synchronize()


def run(label, iters, force_reg):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            if force_reg:
                trainer.iteration = 15
            trainer.train_iteration(real)
        torch.cuda.synchronize()
    ours = stock = 0.0
    by = collections.Counter()
    calls = collections.Counter()
    mine = collections.Counter()
    mine_calls = collections.Counter()
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CUDA:
            continue
        t = e.device_time / 1e3 / iters
        if not any(k in e.name for k in STOCK):
            ours += t
            mine[e.name[:90]] += t
            mine_calls[e.name[:90]] += 1.0 / iters
        else:
            stock += t
            by[e.name[:130]] += t
            calls[e.name[:130]] += 1.0 / iters
    print(f"== {label}: this package's kernels {ours:.1f} ms, stock torch / library kernels {stock:.1f} ms "
          f"({100 * stock / (ours + stock):.1f} %) per iteration")
    for name, t in by.most_common(28):
        print(f"   {t:6.2f} ms  x{calls[name]:6.1f}  {name}")
    if os.environ.get("MSG_SHOW_OURS"):
        for name, t in mine.most_common(24):
            print(f"   ours {t:6.2f} ms  x{mine_calls[name]:6.1f}  {name}")


trainer.iteration = 16
run("plain iteration", 4, False)
run("regularised iteration (R1 + path length)", 2, True)
