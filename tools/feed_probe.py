#!/usr/bin/env python3
"""Where the data feed's residual cost comes from (GPU box only): 256^2, batch 16 plain iterations fed (a) from a resident
batch, (b) through DevicePrefetcher from pageable host memory, (c) from page-locked host memory (no staging copy), (d) through
the prefetcher with batches that already live on the device (thread + queue only), (e) by `.to(device)` per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
from multi_stylegan_amd.data import DevicePrefetcher

DEV = "cuda:0"
torch.manual_seed(1)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
tr = m.ModelWrapper(gen, dis, device=DEV)
tr.generator_ema.compute_dtype = torch.bfloat16
host = torch.rand(16, 2, 3, 256, 256)
pinned = host.pin_memory()
resident = host.to(DEV)
n = 12


def timed(feed):
    tr.iteration = 16
    t0 = None
    for k, batch in enumerate(feed):
        if k == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        tr.train_iteration(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


feeds = {"resident": lambda: [resident] * (n + 2),
         "prefetch pageable": lambda: DevicePrefetcher([host] * (n + 2), DEV),
         "prefetch pinned": lambda: DevicePrefetcher([pinned] * (n + 2), DEV),
         "prefetch device (thread only)": lambda: DevicePrefetcher([resident] * (n + 2), DEV),
         ".to(device) per step": lambda: [host] * (n + 2)}
timed(feeds["resident"]())
for rnd in range(3):
    print("  ".join(f"{name}: {timed(make()):.2f}" for name, make in feeds.items()), flush=True)
