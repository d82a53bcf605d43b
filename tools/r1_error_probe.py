#!/usr/bin/env python3
"""Where the bf16 path's error on the R1 double backward comes from (round-5 review item 3b): the discriminator of BASELINE
config 2 at its own size, batch 2, R1 = 0.5 * mean |d D / d image|^2 and its gradient with respect to a set of weights,
on the GPU in

    (a) fp32 storage, exact contractions                  -- the reference here (held to the CPU oracle at 4e-4 by
                                                              tests/test_hip_models.py::test_config2_r1_double_backward_matches_oracle)
    (b) bf16 storage (the benchmarked path)
    (c) fp32 storage with every WEIGHT rounded to bf16 first  -- the share of (b)'s error that is weight rounding: the same
                                                              perturbation at every pixel, so it does not average out
    (d) bf16 storage from the weights of (c), measured against (c) -- what is left: rounding of the stored maps

Norm-wise relative errors per watched parameter.  GPU box:  python tools/r1_error_probe.py
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multi_stylegan_amd as m
from multi_stylegan_amd import loss as product_loss

DEV = "cuda:0"
WATCH = ["encoder_blocks.0.main_mapping.0.weight", "encoder_blocks.0.residual_mapping.weight",
         "encoder_blocks.1.main_mapping.2.weight", "encoder_blocks.2.theta.weight", "encoder_blocks.2.g.weight",
         "encoder_blocks.2.gamma", "downscale_convolutions.1.0.weight", "encoder_blocks.4.main_mapping.0.weight",
         "decoder_blocks.1.o.weight", "decoder_blocks.3.main_mapping.2.weight", "transposed_convolutions.3.1.weight",
         "final_mapping.1.weight", "classification_head.2.weight"]


def run(dis, dtype, real):
    dis.compute_dtype = dtype
    dis.zero_grad(set_to_none=True)
    x = real.clone().requires_grad_(True)
    s, px = dis(x)
    r1 = product_loss.R1Regularization()(s, x, px)
    r1.backward()
    params = dict(dis.named_parameters())
    return float(r1), {n: params[n].grad.detach().float().clone() for n in WATCH}


def report(name, got, ref):
    e_r1 = abs(got[0] - ref[0]) / abs(ref[0])
    errs = {n: float((got[1][n] - ref[1][n]).norm() / ref[1][n].norm()) for n in WATCH}
    print(f"{name:58s} R1 {e_r1:.2e}   grads: max {max(errs.values()):.2e}  median {sorted(errs.values())[len(errs) // 2]:.2e}   "
          + " ".join(f"{v:.1e}" for v in errs.values()))


def main():
    torch.manual_seed(71)
    gen = torch.Generator().manual_seed(72)
    d0 = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    with torch.no_grad():
        for n, p in d0.named_parameters():
            if n.endswith("gamma"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.3)
            elif n.endswith(".bias") and p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
    real = torch.rand(2, 2, 3, 256, 256, generator=gen).to(DEV)
    d_round = copy.deepcopy(d0)
    with torch.no_grad():
        for p in d_round.parameters():
            if p.ndim >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    d0.to(DEV); d_round.to(DEV)
    a = run(d0, torch.float32, real)
    b = run(d0, torch.bfloat16, real)
    c = run(d_round, torch.float32, real)
    d = run(d_round, torch.bfloat16, real)
    print(f"R1 (fp32) = {a[0]:.6g}")
    report("(b) bf16 storage vs (a) fp32", b, a)
    report("(c) fp32 storage, bf16-rounded weights vs (a)", c, a)
    report("(d) bf16 storage vs (c), both from bf16-rounded weights", d, c)


if __name__ == "__main__":
    main()
