#!/usr/bin/env python3
"""Determinism soak (GPU box only): the same N training iterations twice from the same seeds at the benchmark configuration;
every parameter, EMA parameter and Adam moment must be bit-identical at the end.  A counted s_waitcnt that lets a DMA piece land
late, a hazard on m0 or a race on an LDS stage shows up here as a difference long before it shows up as a wrong loss.
    python tools/determinism_soak.py [iterations]"""
import hashlib
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution

n = int(sys.argv[1]) if len(sys.argv) > 1 else 34          # (two regularised iterations: the 16th and the 32nd)
dev = torch.device("cuda", 0)


def digest(trainer):
    h = hashlib.sha256()
    tensors = [p for mod in (trainer.generator, trainer.discriminator, trainer.generator_ema) for p in mod.parameters()]
    for opt in (trainer.generator_optimizer, trainer.discriminator_optimizer):
        for st in opt.state.values() if hasattr(opt, "state") else ():
            tensors += [v for v in st.values() if torch.is_tensor(v)]
    for t in tensors:
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()


def run():
    torch.manual_seed(7)
    random.seed(7)
    np.random.seed(7)          # (style-mixing index: np.random.randint, as in the reference)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    gen.compute_dtype = dis.compute_dtype = torch.bfloat16
    trainer = m.ModelWrapper(gen, dis, device=dev)
    trainer.generator_ema.compute_dtype = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(11)
    for _ in range(n):
        trainer.train_iteration(torch.rand(16, 2, 3, 256, 256, device=dev, generator=g))
    torch.cuda.synchronize()
    return digest(trainer)


a = run()
b = run()
print(f"{n} iterations, run 1 {a[:16]}  run 2 {b[:16]}  ->", "bit-identical" if a == b else "DIFFERENT")
sys.exit(0 if a == b else 1)
