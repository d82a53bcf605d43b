#!/usr/bin/env python3
"""Soak run (GPU box only): N training iterations at the benchmark configuration on random data, every logged loss
finite, parameters and EMA finite at the end.   python tools/soak.py [iterations]"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0)
torch.manual_seed(7)
random.seed(7)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
for it in range(n):
    trainer.train_iteration(torch.rand(16, 2, 3, 256, 256, device=dev))
    if it % 8 == 7 or it == n - 1:
        logs = trainer.pop_logs()
        bad = {k: v for k, v in logs.items() if not all(math.isfinite(x) for x in v)}
        assert not bad, (it, bad)
        print(f"iteration {it + 1}: " + ", ".join(f"{k}={v[-1]:.4f}" for k, v in sorted(logs.items())), flush=True)
for name, mod in (("G", trainer.generator), ("D", trainer.discriminator), ("EMA", trainer.generator_ema)):
    assert all(torch.isfinite(p).all() for p in mod.parameters()), name
print("soak ok")
