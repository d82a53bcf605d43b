#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at the BASELINE sizes (run on the GPU box):  python tools/microbench.py fir conv wgrad act

Prints one line per case: median time over `--reps` launches (HIP events on the launch stream) and the achieved
algorithmic rate (GB/s for the HBM-bound kernels, TFLOP/s for the MFMA ones).
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda:0"


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def bench_fir(args):
    from multi_stylegan_amd.op_static import upfirdn2d
    fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64).to(DEV)
    for dt in (torch.bfloat16, torch.float32):
        for name, shape, up, down, pad in (("blur 512ch 256^2", (args.batch, 512, 256, 256), 1, 1, (2, 1)),
                                           ("blur 512ch 128^2", (args.batch, 512, 128, 128), 1, 1, (2, 1)),
                                           ("blur 128ch 127->128", (args.batch, 128, 127, 127), 1, 1, (2, 2)),
                                           ("up2 256ch 128->256", (args.batch, 256, 128, 128), 2, 1, (2, 1)),
                                           ("down2 256ch 256->128", (args.batch, 256, 256, 256), 1, 2, (1, 1))):
            if args.only and args.only not in name:
                continue
            x = cl(torch.randn(*shape, device=DEV, dtype=dt))
            y = upfirdn2d(x, fir, up=up, down=down, pad=pad)
            nbytes = (x.numel() + y.numel()) * x.element_size()
            t = timeit(lambda: upfirdn2d(x, fir, up=up, down=down, pad=pad), args.reps)
            print(f"fir  {str(dt)[6:]:9s} {name:22s} {t * 1e6:9.1f} us  {nbytes / t / 1e9:8.1f} GB/s  "
                  f"({nbytes / t / 8e12 * 100:.1f}% of 8 TB/s)", flush=True)
            del x, y


def bench_act(args):
    from multi_stylegan_amd.op_static import fused_bias_noise_leaky_relu
    for dt in (torch.bfloat16, torch.float32):
        x = cl(torch.randn(args.batch, 512, 256, 256, device=DEV, dtype=dt)).requires_grad_(True)
        b = torch.randn(512, device=DEV, requires_grad=True)
        w = torch.randn(1, device=DEV, requires_grad=True)
        nz = torch.randn(args.batch, 1, 256, 256, device=DEV)
        y = fused_bias_noise_leaky_relu(x, b, nz, w)
        gy = torch.randn_like(y)
        t = timeit(lambda: fused_bias_noise_leaky_relu(x, b, nz, w), args.reps)
        nbytes = 2 * x.numel() * x.element_size()
        print(f"act fwd {str(dt)[6:]:9s} 512ch 256^2 {t * 1e6:9.1f} us {nbytes / t / 1e9:8.1f} GB/s", flush=True)
        t = timeit(lambda: torch.autograd.grad(y, (x, b, w), gy, retain_graph=True), args.reps)
        nbytes = 3 * x.numel() * x.element_size()
        print(f"act bwd {str(dt)[6:]:9s} 512ch 256^2 {t * 1e6:9.1f} us {nbytes / t / 1e9:8.1f} GB/s", flush=True)
        del x, y, gy


def bench_conv(args):
    from multi_stylegan_amd import conv_ops
    cases = [("3x3 512->512 256^2 per-sample", "conv", 512, 512, 256, 3, 1, 1, True),
             ("3x3 512->512 128^2 per-sample", "conv", 512, 512, 128, 3, 1, 1, True),
             ("3x3 512->512 64^2 per-sample", "conv", 512, 512, 64, 3, 1, 1, True),
             ("up2 512->512 128->256 per-sample", "up2", 512, 512, 128, 2, 1, 0, True),
             ("1x1 512->3 256^2 per-sample", "conv", 512, 3, 256, 1, 1, 0, True),
             ("3x3 128->128 256^2 shared", "conv", 128, 128, 256, 3, 1, 1, False),
             ("3x3 256->256 128^2 shared", "conv", 256, 256, 128, 3, 1, 1, False),
             ("3x3 1024->1024 16^2 shared", "conv", 1024, 1024, 16, 3, 1, 1, False),
             ("3x3 s2 128->128 256->127 shared", "conv", 128, 128, 256, 3, 2, 0, False),
             ("3x3 s2 256->256 128->63 shared", "conv", 256, 256, 128, 3, 2, 0, False),
             ("3x3 s2 384->384 64->31 shared", "conv", 384, 384, 64, 3, 2, 0, False),
             ("3x3 s2 768->768 32->15 shared", "conv", 768, 768, 32, 3, 2, 0, False),
             ("D 3x3 256->128 256^2 shared", "conv", 256, 128, 256, 3, 1, 1, False),
             ("D 3x3 128->256 256^2 shared", "conv", 128, 256, 256, 3, 1, 1, False),
             ("D 3x3 384->256 128^2 shared", "conv", 384, 256, 128, 3, 1, 1, False),
             ("D 3x3 256->384 128^2 shared", "conv", 256, 384, 128, 3, 1, 1, False),
             ("D 3x3 384->384 64^2 shared", "conv", 384, 384, 64, 3, 1, 1, False),
             ("D 1x1 768->384 64^2 shared", "conv", 768, 384, 64, 1, 1, 0, False),
             ("D 3x3 768->768 32^2 shared", "conv", 768, 768, 32, 3, 1, 1, False),
             ("D 3x3 1024->768 32^2 shared", "conv", 1024, 768, 32, 3, 1, 1, False),
             ("D 1x1 256->128 256^2 shared", "conv", 256, 128, 256, 1, 1, 0, False),
             ("D 3x3 6->128 256^2 shared", "conv", 6, 128, 256, 3, 1, 1, False),
             ("P 1x1 128->256 256^2 shared", "conv", 128, 256, 256, 1, 1, 0, False),
             ("P 1x1 256->384 128^2 shared", "conv", 256, 384, 128, 1, 1, 0, False),
             ("P 1x1 384->256 128^2 shared", "conv", 384, 256, 128, 1, 1, 0, False),
             ("P 1x1 512->3 256^2 per-sample", "conv", 512, 3, 256, 1, 1, 0, True),
             ("P 1x1 3->512 256^2 per-sample", "conv", 3, 512, 256, 1, 1, 0, True)]
    for dt in ((torch.bfloat16,) if not args.f32 else (torch.bfloat16, torch.float32)):
        for name, kind, i, o, r, k, s, p, ps in cases:
            if args.only and args.only not in name:
                continue
            b = args.batch
            x = cl(torch.randn(b, i, r, r, device=DEV, dtype=dt))
            w = torch.randn((b, o, i, k, k) if ps else (o, i, k, k), device=DEV) / math.sqrt(i * k * k)
            g = conv_ops.Geometry(kind, k, k, s, p, (r, r), ps)
            wk, ck = conv_ops._relay_fwd(w, dt)
            extra = int(os.environ.get("MSG_BIG_WPITCH", "0"))       # experiment: padded weight-row pitch (big kernel only)
            if extra and kind != "up2":
                flat = wk.reshape(*wk.shape[:-2], -1)
                wk_p = torch.zeros(*flat.shape[:-1], flat.shape[-1] + extra, device=DEV, dtype=dt)
                wk_p[..., :flat.shape[-1]] = flat
                wk = wk_p
            if kind == "up2":
                wk = wk.transpose(-3, -2).reshape(*wk.shape[:-3], 4 * o, 1, ck).contiguous()
                fn = lambda: conv_ops._launch_fprop(x, wk, ck, None, 4 * o, g.x_hw, 1, 1, 1, 0, 1, True, ps, i)
                flops = 2.0 * b * r * r * 4 * o * i
            else:
                fn = lambda: conv_ops._launch_fprop(x, wk, ck, None, o, g.y_hw, k, k, s, p, 1, False, ps, i)
                flops = 2.0 * b * g.y_hw[0] * g.y_hw[1] * o * i * k * k
            y = fn()
            t = timeit(fn, args.reps)
            print(f"fprop {str(dt)[6:]:9s} {name:34s} {t * 1e6:9.1f} us {flops / t / 1e12:8.1f} TFLOP/s", flush=True)
            gy = torch.randn_like(y) if y.is_contiguous(memory_format=torch.channels_last) else cl(torch.randn(y.shape, device=DEV, dtype=dt))
            if "wgrad" in args.what:
                fw = lambda: conv_ops._g_raw(gy, x, o, i, g)
                t = timeit(fw, args.reps)
                print(f"wgrad {str(dt)[6:]:9s} {name:34s} {t * 1e6:9.1f} us {flops / t / 1e12:8.1f} TFLOP/s", flush=True)
            del x, w, wk, y, gy


def bench_mod(args):
    """Per-sample weight materialisation (msg_scale_rows_cols): algorithmic bytes = fp32 base once + B outputs."""
    from multi_stylegan_amd import conv_ops
    for (o, i, taps) in [(512, 512, 9), (256, 256, 9), (128, 128, 9), (512, 512, 1), (64, 64, 9)]:
        name = f"O{o} I{i} T{taps}"
        if args.only and args.only not in name:
            continue
        base = torch.randn(o, taps, i, device=DEV)
        rs, cs = torch.rand(args.batch, o, device=DEV), torch.rand(args.batch, i, device=DEV)
        out = torch.empty(args.batch, o, taps, i, device=DEV, dtype=torch.float32 if args.f32 else torch.bfloat16)
        t = timeit(lambda: conv_ops._scale_rows_cols(base, rs, cs, out, 0.5), args.reps)
        nbytes = base.numel() * 4 + out.numel() * out.element_size()
        print(f"modw  {name:24s} {t * 1e6:9.1f} us {nbytes / t / 1e9:8.1f} GB/s", flush=True)


def bench_lin(args):
    """Few-row fp32 linear family (csrc/linear.hip); 50 back-to-back launches per timing so launch gaps do not count."""
    from multi_stylegan_amd import conv_ops
    for (m, n, k) in [(16, 512, 512), (32, 512, 512), (16, 128, 768), (16, 1024, 512)]:
        x, w = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV)
        gy, b = torch.randn(m, n, device=DEV), torch.randn(n, device=DEV)
        y, gx, gw, gb = torch.empty(m, n, device=DEV), torch.empty(m, k, device=DEV), torch.empty(n, k, device=DEV), torch.empty(n, device=DEV)
        calls = {"fprop": lambda: conv_ops._lin_call("linear_fprop", 0, x, w, b, y, m, n, k, 1.0, 1.0),
                 "dgrad": lambda: conv_ops._lin_call("linear_dgrad", 0, gy, w, gx, m, n, k, 1.0),
                 "wgrad": lambda: conv_ops._lin_call("linear_wgrad", 0, gy, x, gw, gb, m, n, k, 1.0, 1.0)}
        for name, fn in calls.items():
            def many():
                for _ in range(50):
                    fn()
            t = timeit(many, max(3, args.reps // 3)) / 50
            print(f"linear {name} M{m} N{n} K{k}  {t * 1e6:7.2f} us/launch", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="+", choices=["fir", "act", "conv", "wgrad", "mod", "lin"])
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--only", default="", help="substring filter on the case name")
    args = ap.parse_args()
    if "fir" in args.what:
        bench_fir(args)
    if "act" in args.what:
        bench_act(args)
    if "conv" in args.what or "wgrad" in args.what:
        bench_conv(args)
    if "mod" in args.what:
        bench_mod(args)
    if "lin" in args.what:
        bench_lin(args)
