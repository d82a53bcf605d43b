#!/usr/bin/env python3
"""Only REGULARISED training iterations (lazy R1 + path-length: the every-16th iteration) for a kernel profile:

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 $GRAFT_REPO_ROOT/tools/reg_iteration_profile.py

256^2, batch 16, bf16 storage, 4 regularised iterations (the kernel_stats totals / 4 = one regularised iteration)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m                                                   # noqa: E402
from multi_stylegan_amd.config import generator_config_for_resolution            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
lazy = trainer.hyperparameters["lazy_discriminator_regularization"]
real = torch.rand(16, 2, 3, 256, 256, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    trainer.iteration = lazy - 1
    trainer.train_iteration(real)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / n:.1f} ms per regularised iteration (first one includes warm-up)", file=sys.stderr)
