#!/usr/bin/env python3
"""Plain training iterations on the fp32-STORAGE path (the path the 1e-3 gate is held on) for a kernel profile:

    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python3 $GRAFT_REPO_ROOT/tools/fp32_path_profile.py split_bf16x3

argument: exact | split_bf16x3 (conv_ops.fp32_contraction); 256^2, batch 16, 1 warm-up + 3 timed plain iterations."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m                                                   # noqa: E402
from multi_stylegan_amd import conv_ops                                          # noqa: E402
from multi_stylegan_amd.config import generator_config_for_resolution            # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "split_bf16x3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
conv_ops.fp32_contraction.set(mode)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
trainer = m.ModelWrapper(gen, dis, device=dev)
real = torch.rand(batch, 2, 3, 256, 256, device=dev)
trainer.iteration = 16
trainer.train_iteration(real)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    trainer.train_iteration(real)
torch.cuda.synchronize()
print(f"{mode}: {1e3 * (time.perf_counter() - t0) / 3:.1f} ms per plain iteration, batch {batch}", file=sys.stderr)
