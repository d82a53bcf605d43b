#!/usr/bin/env python3
"""bf16 against fp32 over a TRAINING TRAJECTORY (round-5 review item: the benchmarked path is bf16 storage / bf16 MFMA with fp32
accumulation, the reference is fp32 throughout -- README.md:93 -- and single-step tolerances say nothing about training).

Runs of the same adversarial training from the same initial weights on a mid-size configuration (64x64, generator
5 x 128 channels, the reference's discriminator, batch 8), every random input of every iteration drawn from a seeded CPU
generator and handed to the trainer explicitly (model_wrapper.Draws), so that two runs with the same seed see the same data,
latents, crossover layers and noise planes bit for bit.  A run is `<arithmetic>:<draws seed>[:eps]`:

    f32:<s>      fp32 storage, exact fp32 MFMA contractions   (the path the 1e-3 parity gate is held on; the REFERENCE run of
                                                               seed s: distances below are measured from it)
    bf16:<s>     bf16 storage, bf16 MFMA, fp32 accumulation   (the benchmarked path)
    split:<s>    fp32 storage, six-bf16-product contractions  (fp32-rounding-level arithmetic difference)
    f32:<s>:eps  as f32:<s> from initial weights moved by 1e-6 of their size: what CHAOS alone does to the trajectory, in
                 exact arithmetic (a GAN trajectory amplifies any difference; Adam with beta1 = 0 turns a flipped sign of a
                 rounding-noise gradient into a +-lr step)
    two `f32` runs with different seeds give the run-to-run noise of two fp32 seeds.

Reported: every logged loss of every iteration (JSON), their means over iteration windows, the path-length mean, and at
iterations 16 / 64 / 256 / last the distance of each run's parameters from run A's relative to how far A has moved from
the initial weights, plus the cosine between the two movements.

    python tools/trajectory_ab.py --iterations 512 --out gpurun_out/r05_trajectory      (GPU box)
"""
import argparse
import copy
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--iterations", type=int, default=512)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--resolution", type=int, default=64)
ap.add_argument("--width", type=int, default=128)
ap.add_argument("--out", default="gpurun_out/r05_trajectory")
ap.add_argument("--runs", default="f32:1,bf16:1,split:1,f32:1:eps,f32:2,bf16:2,f32:2:eps,f32:3,bf16:3")
args = ap.parse_args()

import multi_stylegan_amd as m
from multi_stylegan_amd import conv_ops
from multi_stylegan_amd.config import generator_config_for_resolution

DEV = torch.device("cuda", 0)
CHECKPOINTS = sorted({16, 64, 256, args.iterations} & set(range(1, args.iterations + 1)))
G_CFG = dict(generator_config_for_resolution(args.resolution, width=args.width), latent_dimensions=128)
LEVELS = len(G_CFG["channels"]) - 1


def dataset(n_batches=32):
    """Smooth random fields in [0, 1] (a low-resolution random grid, bicubically enlarged): structure a discriminator can
    learn, unlike the benchmark's uniform noise.  Same tensors for every run."""
    g = torch.Generator().manual_seed(777)
    low = torch.rand(n_batches * args.batch, 6, 6, 6, generator=g)
    full = torch.nn.functional.interpolate(low, size=(args.resolution, args.resolution), mode="bicubic", align_corners=False)
    return full.clamp(0, 1).view(n_batches, args.batch, 2, 3, args.resolution, args.resolution)


def draws_for(seed, it, lazy):
    """All random inputs of iteration `it` (1-based) from (seed, it): independent of the compute dtype by construction."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + it)
    rnd = random.Random(seed * 7919 + it)
    b, ld = args.batch, G_CFG["latent_dimensions"]

    def latents(n):
        if rnd.random() < 0.9:                                           # p_mixed_noise (config.py:32)
            return [torch.randn(n, ld, generator=g), torch.randn(n, ld, generator=g)], rnd.randrange(1, 2 * LEVELS + 1)
        return torch.randn(n, ld, generator=g), None

    def planes(n):
        return [torch.randn(n, 1, 4, 4, generator=g)] + \
               [torch.randn(n, 1, 2 ** (i // 2 + 3), 2 ** (i // 2 + 3), generator=g) for i in range(2 * LEVELS)]

    z_d, i_d = latents(b)
    z_g, i_g = latents(b)
    d = dict(z_d=z_d, inject_d=i_d, noise_d=planes(b), z_g=z_g, inject_g=i_g, noise_g=planes(b), cut_mix=False)
    if it % lazy == 0:
        n = max(1, b // 2)
        z_pl, i_pl = latents(n)
        d.update(z_pl=z_pl, inject_pl=i_pl, noise_pl=planes(n),
                 pl_image_noise=torch.randn(n, 2, 3, args.resolution, args.resolution, generator=g))
    return m.Draws(**d).to(DEV)


def flat(module):
    return torch.cat([p.detach().float().flatten() for p in module.parameters()])


def run(tag, init, data):
    arith, seed, *rest = tag.split(":")
    seed = int(seed)
    dtype = torch.bfloat16 if arith == "bf16" else torch.float32
    mode = "split_bf16x3" if arith == "split" else "exact"
    gen, dis = m.MultiStyleGANGenerator(G_CFG), m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    gen.load_state_dict(init[0]); dis.load_state_dict(init[1])
    if rest == ["eps"]:
        pg = torch.Generator().manual_seed(31337)
        with torch.no_grad():
            for p_ in list(gen.parameters()) + list(dis.parameters()):
                p_.mul_(1.0 + 1e-6 * torch.randn(p_.shape, generator=pg))
    gen.compute_dtype = dis.compute_dtype = dtype
    conv_ops.fp32_contraction.set(mode)
    trainer = m.ModelWrapper(gen, dis, device=DEV)
    trainer.generator_ema.compute_dtype = dtype
    lazy = trainer.hyperparameters["lazy_discriminator_regularization"]
    order = random.Random(1000 + seed)
    snaps, pl_mean = {}, {}
    t0 = time.time()
    for it in range(1, args.iterations + 1):
        real = data[order.randrange(data.shape[0])].to(DEV)
        trainer.train_iteration(real, draws_for(seed, it, lazy))
        if it in CHECKPOINTS:
            snaps[it] = (flat(trainer.generator).cpu(), flat(trainer.discriminator).cpu(), flat(trainer.generator_ema).cpu())
            pl_mean[it] = float(trainer.path_length_regularization.mean_path_length.reshape(-1)[0])
    torch.cuda.synchronize()
    took = time.time() - t0
    logs = trainer.pop_logs()
    assert all(v == v and abs(v) != float("inf") for vs in logs.values() for v in vs), f"run {tag}: non-finite loss"
    # the EMA generator's images for fixed latents, all runs evaluated in fp32 storage
    trainer.generator_ema.compute_dtype = torch.float32
    conv_ops.fp32_contraction.set("exact")
    g = torch.Generator().manual_seed(4242)
    z = torch.randn(16, G_CFG["latent_dimensions"], generator=g).to(DEV)
    noise = [torch.randn(16, 1, 4, 4, generator=g).to(DEV)] + \
            [torch.randn(16, 1, 2 ** (i // 2 + 3), 2 ** (i // 2 + 3), generator=g).to(DEV) for i in range(2 * LEVELS)]
    with torch.no_grad():
        images = trainer.generator_ema(input=z, noise=noise).float().cpu()
    print(f"run {tag}: {args.iterations} iterations in {took:.1f} s ({1e3 * took / args.iterations:.1f} ms / iteration)", flush=True)
    return {"logs": logs, "snaps": snaps, "pl_mean": pl_mean, "images": images, "seconds": took}


def main():
    torch.manual_seed(99)
    g0 = m.MultiStyleGANGenerator(G_CFG)
    d0 = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    init = (copy.deepcopy(g0.state_dict()), copy.deepcopy(d0.state_dict()))
    theta0 = (flat(g0), flat(d0), flat(g0))
    data = dataset()
    tags = args.runs.split(",")
    res = {t: run(t, init, data) for t in tags}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({t: {k: [float(f"{v:.6g}") for v in vs] for k, vs in r["logs"].items()} for t, r in res.items()},
              open(args.out + "_losses.json", "w"))
    names = {"f32": "fp32 exact", "bf16": "bf16", "split": "fp32 six-product"}
    label = lambda t: names[t.split(":")[0]] + ", seed " + t.split(":")[1] + (", initial weights x (1 + 1e-6 N(0,1))" if t.endswith(":eps") else "")
    md = [f"# bf16 against fp32 over {args.iterations} training iterations (tools/trajectory_ab.py)", "",
          f"{args.resolution}x{args.resolution}, generator {len(G_CFG['channels'])} x {args.width} channels (latent "
          f"{G_CFG['latent_dimensions']}), the reference's discriminator configuration, batch {args.batch}; identical initial "
          f"weights; every random input of an iteration drawn from (seed, iteration) on the CPU and handed to the trainer.", "",
          "| run | what | ms / iteration |", "|---|---|---|"]
    md += [f"| {t} | {label(t)} | {1e3 * res[t]['seconds'] / args.iterations:.1f} |" for t in tags]
    # ---- loss windows
    edges = [0] + CHECKPOINTS
    keys = list(res[tags[0]]["logs"])
    every = {"loss_discriminator_regularization", "path_length", "loss_path_length_regularization"}
    md += ["", "## Logged losses, mean over iteration windows", "",
           "(per-iteration values of every run: `" + os.path.basename(args.out) + "_losses.json`; the lazy terms are logged every "
           "16th iteration)", "", "| loss | window | " + " | ".join(tags) + " |", "|---|---|" + "---|" * len(tags)]
    for k in keys:
        for lo, hi in zip(edges[:-1], edges[1:]):
            row = []
            for t in tags:
                vs = res[t]["logs"].get(k, [])
                sel = vs[lo // 16:hi // 16] if k in every else vs[lo:hi]
                row.append(f"{sum(sel) / len(sel):.4g}" if sel else "-")
            md.append(f"| {k} | {lo + 1}-{hi} | " + " | ".join(row) + " |")
    # ---- parameter drift, each run against the exact-fp32 run of ITS seed; the fp32 runs of other seeds against the first one
    refs = [t for t in tags if t.startswith("f32:") and not t.endswith(":eps")]
    md += ["", "## Parameter distance from the exact-fp32 run of the same seed, relative to that run's own movement from the initial weights", "",
           "`|theta_X(t) - theta_R(t)| / |theta_R(t) - theta(0)|` (and the cosine between the two movements) for the generator, the "
           "discriminator and the EMA generator; an exact-fp32 run of ANOTHER seed is measured from the first fp32 run (seed noise)", "",
           "| iteration | run | from | G | D | G_ema | path-length mean |", "|---|---|---|---|---|---|---|"]
    for it in CHECKPOINTS:
        for t in tags:
            arith, seed = t.split(":")[:2]
            ref = f"f32:{seed}"
            if t == ref:
                a = res[t]["snaps"][it]
                moved = ", ".join(f"{(x - x0).norm() / x0.norm():.3e}" for x, x0 in zip(a, theta0))
                if t == refs[0] or not refs:
                    md.append(f"| {it} | {t} | - | moved {moved} of its norm | | | {res[t]['pl_mean'][it]:.4g} |")
                    continue
                ref = refs[0]
            if ref not in res:
                continue
            cells = []
            for x, xa, x0 in zip(res[t]["snaps"][it], res[ref]["snaps"][it], theta0):
                move_a, move_x = xa - x0, x - x0
                cos = float(torch.dot(move_a, move_x) / (move_a.norm() * move_x.norm() + 1e-30))
                cells.append(f"{float((x - xa).norm() / (move_a.norm() + 1e-30)):.3f} (cos {cos:.3f})")
            md.append(f"| {it} | {t} | {ref} | " + " | ".join(cells) + f" | {res[t]['pl_mean'][it]:.4g} |")
    md += ["", "## EMA generator images for 16 fixed latents / noise planes (evaluated in fp32)", "",
           "| pair | RMS difference | relative to the RMS of the reference run's images |", "|---|---|---|"]
    for t in tags:
        arith, seed = t.split(":")[:2]
        ref = f"f32:{seed}" if t != f"f32:{seed}" else (refs[0] if refs and t != refs[0] else None)
        if ref and ref in res:
            d = float((res[t]["images"] - res[ref]["images"]).square().mean().sqrt())
            md.append(f"| {t} vs {ref} | {d:.4g} | {d / float(res[ref]['images'].square().mean().sqrt()):.3f} |")
    open(args.out + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
