#!/usr/bin/env python3
"""Benchmark of the Multi-StyleGAN G+D training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full adversarial training iteration (D step, G step, EMA, lazy R1 / path-length regularisers at
their natural 1-in-16 cadence) on a synthetic batch that is already resident in HBM.  Default workload =
BASELINE.json configs[1]: 256x256, seq 3 x 2 channels, batch 16 per GPU, bf16 storage / fp32 accumulate, fp32
master weights.  Weak scaling: every rank runs the same per-GPU batch; gradients are averaged over RCCL.

Warm-up: W untimed iterations, the last of them a regularised one (so that every code path has run once before the
timed region; the timed K iterations fire the regularisers at their natural cadence).

`value`: training runs 15 plain iterations per regularised one.  A window of K iterations holds floor(K/16) or one more
regularised iteration, so its raw average is optimistic or pessimistic depending on K and the phase.  The timed region
therefore also clocks every iteration (one HIP event per iteration boundary) and `value` is the 15:1 amortised rate
    world * batch / ((15 * plain_ms + regularised_ms) / 16)
from the timed region's own plain and regularised iterations (`timed_region` carries both and the raw average; when the
window holds no regularised iteration or no plain one, `value` falls back to the raw average and says so).

Rank 0 prints ONE JSON line; besides the contract's keys it carries
  roofline        the dominant hand-written kernel, timed per launch with HIP events inside the timed region
  cpu_baseline    the CPU oracle (oracle/) timed on this host on a bounded sample of the workload (64^2, batch 4, and one
                  iteration at the benchmark's own resolution for a same-resolution ratio)
  value_fp32_path the same iteration on the fp32-storage path (the path the 1e-3 parity gate is held on), batch 4
  losses          the last value of every logged loss (asserted finite)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--elide-dead-work", action="store_true",
                    help="skip work whose results the reference discards (dead 2nd-stream convs, D weight grads in "
                         "the G step); identical training trajectory, not used for the headline value")
    ap.add_argument("--ada", action="store_true",
                    help="wrap the discriminator in adaptive discriminator augmentation (BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-clock", action="store_true")
    ap.add_argument("--clock-all", action="store_true",
                    help="HIP-event timing of EVERY launch (per-kernel table in the JSON line); by default only the two "
                         "roofline kernels are timed, which keeps the event overhead out of the host path")
    ap.add_argument("--clock-only", choices=["all", "plain", "regularised"], default="all",
                    help="restrict the per-kernel clock to the plain / the regularised (R1 + path length) iterations "
                         "(diagnostic; tools/shape_table.py divides by `clock_iterations`)")
    ap.add_argument("--cpu-baseline-iters", type=int, default=4)
    ap.add_argument("--no-cpu-same-resolution", action="store_true",
                    help="skip the CPU oracle's one iteration at the benchmark's own resolution (about a minute at 256^2)")
    ap.add_argument("--no-h2d-leg", action="store_true",
                    help="skip the leg that feeds the step from pageable host memory through the prefetcher")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-storage leg (value_fp32_path)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device): rehearses the "
                         "launcher, the rendezvous and the exchange path on a 1-GPU box; not a measurement")
    return ap.parse_args()


def baseline_config(args, world) -> str:
    """Which BASELINE.json config this run is, if any."""
    if args.dtype != "bf16":
        return ""
    if args.ada:
        return " + ADA (BASELINE configs[4])" if args.resolution == 256 and world == 8 else " + ADA"
    if args.resolution == 256 and args.batch == 16:
        return " (BASELINE configs[1])" if world == 1 else (" (BASELINE configs[2])" if world == 8 else
                                                            " (BASELINE configs[1] per GPU)")
    if args.resolution == 512 and args.batch == 8:
        return " (BASELINE configs[3] per GPU)"
    return ""


def _is_flops(key: str) -> bool:
    """Kernel-clock keys whose `work` is FLOPs (the contractions); everything else counts bytes."""
    return key.startswith(("conv", "linear", "nl_attention"))


def note(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """CPU threads this process may really use (affinity mask, and the GPU box's 16-core share per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("MSG_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(iters: int, same_resolution: int = 0):
    """The CPU oracle on BASELINE config 1 (64x64, 5 x 512 channels, B=4): `iters` plain iterations after one
    warm-up.  A bounded sample: the 256^2 workload itself takes minutes per iteration on a CPU."""
    from multi_stylegan_amd.config import generator_config_for_resolution
    from oracle import models as om, train as ot
    import copy
    threads = host_threads()
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    g, d = om.Generator(generator_config_for_resolution(64)), om.Discriminator(no_rfp=True)
    g_ema = copy.deepcopy(g)
    og, od = ot.make_optimizers(g, d)
    pl = ot.PathLength()
    real = torch.rand(4, 2, 3, 64, 64)
    ot.train_iteration(g, d, g_ema, og, od, pl, real, 1)
    t0 = time.perf_counter()
    for it in range(iters):
        ot.train_iteration(g, d, g_ema, og, od, pl, real, 2 + it)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    ot.train_iteration(g, d, g_ema, og, od, pl, real, 16)            # R1 + path-length regularisers fire
    dt_lazy = time.perf_counter() - t1
    amortised = 16 * 4 / (15 * dt / iters + dt_lazy)                 # 15 plain + 1 regularised iteration
    out = {"value": round(amortised, 4), "unit": "img/s", "cores": threads, "kind": "port",
           "plain_iteration_s": round(dt / iters, 3), "regularised_iteration_s": round(dt_lazy, 3),
           "sample": f"{iters} plain training iterations + 1 iteration with the lazy R1 / path-length regularisers "
                     f"of the CPU oracle at 64x64, batch 4 (BASELINE config 1), weighted 15:1 as in training; "
                     f"torch {torch.__version__}, {threads} threads"}
    if same_resolution:
        # one PLAIN iteration at the benchmark's own resolution and channel configuration ("optionally one 256^2
        # B=2 iteration for a same-resolution ratio"): bounded at batch 2, no warm-up (an iteration takes minutes)
        del g, d, g_ema, og, od
        res, bsz = same_resolution, 2
        torch.manual_seed(4321)
        g, d = om.Generator(generator_config_for_resolution(res)), om.Discriminator(no_rfp=True)
        g_ema = copy.deepcopy(g)
        og, od = ot.make_optimizers(g, d)
        real = torch.rand(bsz, 2, 3, res, res)
        t2 = time.perf_counter()
        ot.train_iteration(g, d, g_ema, og, od, ot.PathLength(), real, 2)
        dt_same = time.perf_counter() - t2
        out["same_resolution"] = {"value": round(bsz / dt_same, 4), "unit": "img/s", "iteration_s": round(dt_same, 2),
                                  "sample": f"1 plain training iteration of the CPU oracle at {res}x{res}, batch {bsz}, "
                                            f"no warm-up, {threads} threads"}
    return out


def fp32_leg(args, dev, batch: int = 4, steps: int = 3, split: str = ""):
    """The same training iteration on the fp32-storage path -- exact-fp32 MFMA contractions, the path the 1e-3 parity
    gate of tests/test_hip_models.py is held on -- at the configuration's own batch: one regularised warm-up iteration, `steps`
    timed plain iterations and one timed regularised iteration, amortised 15:1 like `value`."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd import conv_ops
    from multi_stylegan_amd.config import generator_config_for_resolution
    if split:
        # `split`: the contractions as six bf16 MFMA products on (hi, mid, lo) splits of the fp32 operands (MSG_F32_SPLIT; held to
        # the same step-trace tolerances as the exact path: tests/test_hip_models.py::test_train_iteration_split_bf16_products)
        with conv_ops.fp32_contraction(split):
            out = fp32_leg(args, dev, batch, steps)
        out["dtype"] = "f32 storage, six bf16 MFMA products per product on (hi, mid, lo) splits (all 24 mantissa bits), fp32 accumulate"
        out["sample"] = out["sample"].replace("exact-fp32 MFMA", "six bf16 MFMA products per product")
        out["step_parity"] = "gradients 1e-3 / norm 1e-4 / movement 2e-3, as the exact path (tests/test_hip_models.py)"
        return out
    torch.manual_seed(1234)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(args.resolution))
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    trainer = m.ModelWrapper(gen, dis, device=dev)
    lazy = trainer.hyperparameters["lazy_discriminator_regularization"]
    real = torch.rand(batch, 2, 3, args.resolution, args.resolution, device=dev)
    trainer.iteration = lazy - 1
    trainer.train_iteration(real)                              # warm-up, regularised: iteration 16
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_iteration(real)                          # 17, 18, ...: plain
    torch.cuda.synchronize(dev)
    plain_ms = 1e3 * (time.perf_counter() - t0) / steps
    trainer.iteration = 2 * lazy - 1
    t1 = time.perf_counter()
    trainer.train_iteration(real)                              # 32: regularised
    torch.cuda.synchronize(dev)
    reg_ms = 1e3 * (time.perf_counter() - t1)
    last = {k: v[-1] for k, v in trainer.pop_logs().items()}
    assert all(v == v and abs(v) != float("inf") for v in last.values()), f"non-finite loss on the fp32 path: {last}"
    step_ms = ((lazy - 1) * plain_ms + reg_ms) / lazy
    return {"value": round(batch / (step_ms * 1e-3), 3), "unit": "img/s", "dtype": "f32", "batch": batch,
            "plain_ms": round(plain_ms, 2), "regularised_ms": round(reg_ms, 2), "ms_per_step": round(step_ms, 2),
            "sample": f"{steps} plain + 1 regularised iteration at {args.resolution}x{args.resolution}, batch {batch}, "
                      f"fp32 storage and exact-fp32 MFMA, amortised {lazy - 1}:1"}


def elided_leg(args, dev, dtype, steps: int = 6):
    """The same iteration without the two pieces of work whose results the reference computes and throws away -- the second
    stream's main convolutions and the discriminator's weight gradients in the generator step (Q11).  Identical
    training trajectory, bit for bit (tested); reported beside `value`, which executes both like the reference does."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd.config import generator_config_for_resolution
    torch.manual_seed(1234)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(args.resolution))
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    gen.compute_dtype = dis.compute_dtype = dtype
    gen.elide_dead_branch = True
    trainer = m.ModelWrapper(gen, dis, device=dev, skip_discriminator_weight_grads_in_generator_step=True)
    trainer.generator_ema.compute_dtype = dtype
    lazy = trainer.hyperparameters["lazy_discriminator_regularization"]
    real = torch.rand(args.batch, 2, 3, args.resolution, args.resolution, device=dev)
    trainer.iteration = lazy - 2
    for _ in range(2):
        trainer.train_iteration(real)                          # warm-up: 15 (plain), 16 (regularised)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_iteration(real)                          # 17 ...: plain
    torch.cuda.synchronize(dev)
    plain_ms = 1e3 * (time.perf_counter() - t0) / steps
    trainer.iteration = 2 * lazy - 1
    t1 = time.perf_counter()
    trainer.train_iteration(real)                              # 32: regularised
    torch.cuda.synchronize(dev)
    reg_ms = 1e3 * (time.perf_counter() - t1)
    step_ms = ((lazy - 1) * plain_ms + reg_ms) / lazy
    return {"value": round(args.batch / (step_ms * 1e-3), 3), "unit": "img/s", "plain_ms": round(plain_ms, 2),
            "regularised_ms": round(reg_ms, 2), "ms_per_step": round(step_ms, 2),
            "sample": f"{steps} plain + 1 regularised iteration with --elide-dead-work (dead second-stream convolutions and the "
                      f"discriminator's weight gradients in the generator step skipped; same trajectory), amortised {lazy - 1}:1"}


def multi_rank_fields(world, backend, rehearsal, batch, steps, per_rank_seconds, overlap_off_ms):
    """What the JSON line says about the ranks: how many exchanged gradients and over which backend ("nccl" IS RCCL on ROCm;
    "gloo" only in the one-GPU rehearsal), every rank's own rate over the timed region (the job's rate is the slowest
    rank's: `value` uses the MAX of these times), and what overlapping the bucket exchange with backward buys (ms per step
    with the exchange issued from the gradient hooks vs after backward; None on one rank)."""
    elapsed = max(per_rank_seconds)
    return {"rccl_ranks": world, "rehearsal_shared_gpu": bool(rehearsal), "backend": backend if world > 1 else None,
            "per_rank_img_per_s": [round(batch * steps / t, 2) for t in per_rank_seconds],
            "overlap": {"on_ms_per_step": round(1e3 * elapsed / steps, 2), "off_ms_per_step": round(overlap_off_ms, 2)}
            if overlap_off_ms is not None else None}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one
    process per GPU, RCCL rendezvous on 127.0.0.1) from a parent that never touches the GPU, relay their output
    (rank 0 prints the JSON line) and return the launcher's exit code."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    if visible < n and "--rehearse-on-one-gpu" not in sys.argv:
        print(f"bench.py: --gpus {n} but only {visible} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:                       # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"no launcher environment: starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    from multi_stylegan_amd import dist as msg_dist
    rank, world, local_rank = msg_dist.init_from_env("gloo" if args.rehearse_on_one_gpu else None)
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (the product path has no CPU fallback)"
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import multi_stylegan_amd as m
    from multi_stylegan_amd import _lib
    from multi_stylegan_amd.config import generator_config_for_resolution

    torch.manual_seed(1234)                                   # same init on every rank (and broadcast anyway)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(args.resolution))
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    gen.compute_dtype = dis.compute_dtype = dtype
    gen.elide_dead_branch = args.elide_dead_work
    if args.ada:
        dis = m.AdaptiveDiscriminatorAugmentation(dis)
    trainer = m.ModelWrapper(gen, dis, device=dev,
                             skip_discriminator_weight_grads_in_generator_step=args.elide_dead_work)
    trainer.generator_ema.compute_dtype = dtype
    torch.manual_seed(1234 + rank)                            # different data / z / noise per rank
    import random
    random.seed(1234 + rank)
    import numpy
    numpy.random.seed(1234 + rank)                            # the style-mixing crossover layer is drawn with numpy
    real = torch.rand(args.batch, 2, 3, args.resolution, args.resolution, device=dev)

    hp_lazy = trainer.hyperparameters["lazy_discriminator_regularization"]

    # ADA augments the batch it is given IN PLACE (as the reference does, adaptive_discriminator_augmentation.py:64-68);
    # a data loader hands out a fresh batch every step, so the resident synthetic batch is cloned per step (25 MB) then
    batch_of = (lambda: real.clone()) if args.ada else (lambda: real)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    note(f"models built on {dev}; warm-up ({args.warmup} iterations, includes library kernel selection)")
    for i in range(args.warmup):
        if i == args.warmup - 1:
            # the last warm-up iteration is a REGULARISED one (R1 + path length, the every-16th-iteration work): their
            # second-order kernels and library GEMM shapes are otherwise first met -- one-time kernel selection, hundreds
            # of ms -- inside the timed region, where the regularisers still fire at their natural 1-in-16 cadence
            count = trainer.iteration
            trainer.iteration = hp_lazy - 1
            trainer.train_iteration(batch_of())
            trainer.iteration = count + 1
        else:
            trainer.train_iteration(batch_of())
        torch.cuda.synchronize(dev)
        note(f"warm-up iteration {i + 1} done")
    trainer.pop_logs()
    clock_all = args.clock_all or bool(int(os.environ.get("MSG_CLOCK_SHAPES", "0")))
    roofline_keys = (("conv_fprop_row3", "conv_fprop_pp") if args.dtype == "bf16" else ("conv_fprop_dma",)) + ("upfirdn2d",)
    _lib.kernel_clock.reset(enabled=not args.no_kernel_clock, only=None if clock_all else roofline_keys)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    first_iteration = trainer.iteration + 1
    t0 = time.perf_counter()
    marks[0].record()
    clock_iterations = 0
    for k in range(args.steps):
        if args.clock_only != "all" and not args.no_kernel_clock:
            _lib.kernel_clock.enabled = ((first_iteration + k) % hp_lazy == 0) == (args.clock_only == "regularised")
        clock_iterations += int(_lib.kernel_clock.enabled)
        trainer.train_iteration(batch_of())
        marks[k + 1].record()                                  # (one event per iteration: the plain / regularised split)
    barrier()
    elapsed = time.perf_counter() - t0
    iter_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    is_reg = [(first_iteration + k) % hp_lazy == 0 for k in range(args.steps)]
    plain = [t for t, r in zip(iter_ms, is_reg) if not r]
    regd = [t for t, r in zip(iter_ms, is_reg) if r]
    split = torch.tensor([sum(plain) / max(1, len(plain)), sum(regd) / max(1, len(regd))], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(split, op=torch.distributed.ReduceOp.MAX)     # the job is as slow as its slowest rank
    plain_ms, reg_ms = split.tolist()
    per_rank = [elapsed]
    overlap_off_ms = None
    if world > 1:
        every = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        torch.distributed.all_gather(every, torch.tensor([elapsed], device=dev, dtype=torch.float64))
        per_rank = [t.item() for t in every]
        elapsed = max(per_rank)                            # the job is as slow as its slowest rank
        # what the overlap buys: the same steps with every bucket exchanged after backward instead of during it
        # (outside the timed region of `value`)
        n_off = min(args.steps, 8)
        for red in (trainer.generator_reducer, trainer.discriminator_reducer):
            red.overlap = False
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_off):
            trainer.train_iteration(batch_of())
        barrier()
        t_off = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t_off, op=torch.distributed.ReduceOp.MAX)
        overlap_off_ms = 1e3 * t_off.item() / n_off
    note(f"timed region done: {elapsed:.2f} s for {args.steps} steps")
    clock = _lib.kernel_clock.summary()
    _lib.kernel_clock.reset(enabled=False)
    h2d = None
    if world == 1 and not args.no_h2d_leg and plain:
        # the same plain iterations with the real batch arriving from PAGEABLE host memory every step through the data
        # feed (multi_stylegan_amd.data.DevicePrefetcher: pinned staging + copy stream, one iteration ahead) -- the only
        # host->device traffic of the reference's loop (model_wrapper.py:253-256).  Outside `value`'s timed region.
        from multi_stylegan_amd.data import DevicePrefetcher
        host_batch = real.cpu()
        n_h2d = 8
        trainer.iteration = 4 * hp_lazy                        # the next n_h2d + 2 (< 16) iterations are plain ones
        t1 = None
        for k, batch in enumerate(DevicePrefetcher([host_batch] * (n_h2d + 2), dev)):
            if k == 2:                                         # (the feed's start-up -- its thread, the first page-locked
                barrier()                                      #  staging buffers -- happens once per epoch, not per step)
                t1 = time.perf_counter()
            trainer.train_iteration(batch)
        barrier()
        h2d_ms = 1e3 * (time.perf_counter() - t1) / n_h2d
        h2d = {"plain_ms_with_h2d": round(h2d_ms, 2), "plain_ms_resident": round(plain_ms, 2),
               "bytes_per_step": host_batch.numel() * host_batch.element_size(),
               "img_per_s_with_h2d": round(args.batch / (((hp_lazy - 1) * h2d_ms + (reg_ms if regd else h2d_ms)) / hp_lazy * 1e-3), 3),
               "sample": f"{n_h2d} plain iterations, the real batch copied from pageable host memory every step through "
                         f"data.DevicePrefetcher (pinned double buffer, copy stream); amortised with the resident "
                         f"regularised iteration"}
    logs = trainer.pop_logs()
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2 ** 30

    if rank == 0:
        raw_ms = 1e3 * elapsed / args.steps
        amortise = bool(plain) and bool(regd)
        step_ms = ((hp_lazy - 1) * plain_ms + reg_ms) / hp_lazy if amortise else raw_ms
        value = world * args.batch / (step_ms * 1e-3)
        # dominant kernel (rocprofv3: ~44 % of GPU time): the bf16 implicit-GEMM conv on the matrix cores.
        # achieved = algorithmic FLOPs of its launches inside the timed region / their HIP-event durations.
        pmc, pmc_file = {}, None       # HBM-side bytes per launch from separate rocprofv3 --pmc passes of this workload
        for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")),
                           reverse=True):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
                pmc_file = name
                break
            except (OSError, KeyError, ValueError):
                continue

        def leg(key, bound, peak, unit, name):
            if key not in clock:
                return None
            c = clock[key]
            rate = c["work"] / (c["total_ms"] * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
            return {"kernel": name, "bound": bound, "achieved": round(rate, 1), "peak": peak, "unit": unit,
                    "frac": round(rate / peak, 4),
                    "traffic": (pmc.get(key) or pmc.get(key.replace("+act", ""), {})).get("traffic_bytes_per_launch")
                    if args.batch == 16 and
                    args.resolution == 256 and args.dtype == "bf16" else None,
                    "traffic_unit": f"HBM-side bytes per launch, rocprofv3 PMC (profiles/{pmc_file})",
                    "launches": c["launches"],
                    "avg_us": round(c["avg_us"], 2), "algorithmic_work_per_launch": round(c["work"] / c["launches"])}
        mf = args.dtype == "bf16"
        # (bf16: whichever of the two large-tile conv kernels spent more time in the timed region is the dominant one)
        names = {"conv_fprop_row3": "conv_fprop_row3_kernel<4,4,true> (implicit-GEMM 3x3 conv fwd + data-grad, 256x256 tile, activation "
                                    "tile shared by the three horizontal taps, MFMA 16x16x32 bf16; rocprofv3 symbol "
                                    "conv_fprop_row3_kernel<4, 4, true, 0>)",
                 "conv_fprop_pp": "conv_fprop_pp_kernel (implicit-GEMM conv fwd + data-grad, 256x256 ping-pong tile, MFMA "
                                  "32x32x16 bf16)"}
        dom = max(names, key=lambda k: clock.get(f"{k}/{args.dtype}", {}).get("total_ms", 0.0))
        dom_key = f"{dom}/{args.dtype}"
        roof = leg(dom_key, "mfma", MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", names[dom]) \
            if mf else leg(f"conv_fprop_dma/{args.dtype}", "mfma", MFMA_F32_PEAK_TFLOPS, "TFLOP/s",
                           "conv_fprop_kernel<float, true> (implicit-GEMM conv, MFMA 32x32x2 f32)")
        # the same tile doing the same contraction with an activation backward in its epilogue: its own instantiation (rocprofv3
        # symbol conv_fprop_row3_kernel<4, 4, true, 1>), reported beside the dominant kernel, not inside its average
        roof_twin = leg(f"conv_fprop_row3_actbwd/{args.dtype}", "mfma", MFMA_BF16_PEAK_TFLOPS, "TFLOP/s",
                        "conv_fprop_row3_kernel<4,4,true,1> (the dominant kernel's contraction + the preceding activation's "
                        "backward and its bias / noise-weight partial sums in the epilogue)") if mf else None
        # the FIR launches that carry the bytes: the blur behind every upsampling styled conv, which also applies that
        # layer's noise + bias + leaky ReLU (algorithmic bytes: input + output + noise plane)
        roof_fir = leg(f"upfirdn2d/{args.dtype}/up1down1/sep+act", "hbm", HBM_PEAK_GBS, "GB/s",
                       "blur_sep_kernel (4x4 FIR blur up=down=1 + fused noise/bias/leaky-ReLU, separable sliding window, "
                       "channels-last)") or \
            leg(f"upfirdn2d/{args.dtype}/up1down1/sep", "hbm", HBM_PEAK_GBS, "GB/s",
                "blur_sep_kernel (4x4 FIR blur up=down=1, separable sliding window, channels-last)") or \
            leg(f"upfirdn2d/{args.dtype}/up1down1/vec", "hbm", HBM_PEAK_GBS, "GB/s",
                "upfirdn2d_vec_kernel<up=1,down=1> (4x4 FIR blur, channels-last)")
        kernels = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                       ("TFLOP/s" if _is_flops(k) else "GB/s"):
                           round(v["work"] / (v["total_ms"] * 1e-3) / (1e12 if _is_flops(k) else 1e9), 1)}
                   for k, v in sorted(clock.items())}
        last = {k: v[-1] for k, v in logs.items() if v}
        assert last and all(v == v and abs(v) != float("inf") for v in last.values()), f"non-finite loss: {last}"
        out = {
            "metric": "training images/sec (G+D step, 256^2, seq=3x2ch)", "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.resolution}x{args.resolution}, seq_len=3, 2 channels, batch={args.batch}/GPU"
                                   f"{baseline_config(args, world)}; full iteration: D step + G step + EMA, lazy R1 and "
                                   f"path-length every 16th; one synthetic real batch per rank, resident in HBM and "
                                   f"re-used every step (fresh z / noise per step)",
                       "global_batch": world * args.batch,
                       "parallelism": f"dp{world}", "dead_work_elided": bool(args.elide_dead_work),
                       "ada": bool(args.ada)},
            **multi_rank_fields(world, torch.distributed.get_backend() if world > 1 else None, args.rehearse_on_one_gpu,
                                args.batch, args.steps, per_rank, overlap_off_ms),
            "timed_region": {"iterations": [first_iteration, first_iteration + args.steps - 1],
                             "regularised_iterations": len(regd), "plain_ms": round(plain_ms, 2) if plain else None,
                             "regularised_ms": round(reg_ms, 2) if regd else None, "raw_ms_per_step": round(raw_ms, 2),
                             "raw_img_per_s": round(world * args.batch * args.steps / elapsed, 3),
                             "value_is": f"{hp_lazy - 1}:1 amortised (plain : regularised iterations, as in training)"
                             if amortise else "raw average of the window (it holds no regularised or no plain iteration)"},
            "roofline": roof, "roofline_upfirdn2d": roof_fir, "roofline_act_backward_twin": roof_twin, "kernels": kernels,
            "clock_iterations": clock_iterations, "peak_mem_GiB": round(peak_mem, 2),
            "losses": {k: round(v, 5) for k, v in last.items()},
            "h2d": h2d,
            # which shared library produced the numbers: the product build unless MSG_LIB_VARIANT named an A/B or diagnostic
            # build (tools only; a line that carries a variant is not the product's)
            "library": {"file": os.path.basename(_lib.LIB_PATH), "abi": int(_lib.lib().msg_abi_version()),
                        "variant": os.environ.get("MSG_LIB_VARIANT") or None,
                        "call_wrappers": "generated (_msg_fastcall)" if _lib.lib().fastcall else "ctypes"},
        }
        if world == 1 and not args.no_fp32_leg and args.dtype == "bf16" and not args.rehearse_on_one_gpu:
            del trainer, gen, dis, real
            torch.cuda.empty_cache()
            if not args.elide_dead_work:
                note("dead-work-elided leg ...")
                out["value_dead_work_elided"] = elided_leg(args, dev, dtype)
                torch.cuda.empty_cache()
            note(f"fp32-storage leg (batch {args.batch}) ...")
            out["value_fp32_path"] = fp32_leg(args, dev, batch=args.batch)
            torch.cuda.empty_cache()
            note(f"fp32-storage legs with split-bf16 products (batch {args.batch}) ...")
            out["value_fp32_split_path"] = fp32_leg(args, dev, batch=args.batch, split="split_bf16x3")
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N=1 only
            note("CPU baseline (oracle, 64x64, B=4" + ("" if args.no_cpu_same_resolution else
                                                       f"; one iteration at {args.resolution}^2, B=2") + ") ...")
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_iters,
                                               0 if args.no_cpu_same_resolution else args.resolution)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
