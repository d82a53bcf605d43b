"""Every backward kernel is deterministic: no float atomics anywhere on the training path, sums that span workgroups are
formed from per-workgroup partials in a fixed order (csrc/conv_wgrad.hip: K-slice slabs + wgrad_reduce_kernel,
csrc/bias_act.hip: part_b / part_n + bias_act_bwd_reduce_kernel).  Identical inputs therefore give bit-identical gradients
and a bit-reproducible training step -- what lets the second-order (R1, path-length) steps be held to the north-star
tolerance in tests/test_hip_models.py instead of to the run-to-run spread of atomic accumulation."""
import copy
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPEATS = 8

# (name, batch, in, out, h, w, k, stride, pad, kind, per_sample, dtype): one shape per weight-gradient kernel / addressing path
WGRAD_CASES = [
    ("row3s_shared_128ch_256px", 4, 128, 128, 256, 256, 3, 1, 1, "conv", False, torch.bfloat16),      # many K-slices, one tile
    ("row3s_shared_512ch_64px", 4, 512, 512, 64, 64, 3, 1, 1, "conv", False, torch.bfloat16),
    ("row3s_w32_shared_768ch", 8, 768, 768, 32, 32, 3, 1, 1, "conv", False, torch.bfloat16),
    ("row3s_per_sample", 3, 256, 256, 64, 64, 3, 1, 1, "conv", True, torch.bfloat16),                  # no split
    ("uni_1x1_shared", 4, 256, 128, 128, 128, 1, 1, 0, "conv", False, torch.bfloat16),
    ("uni_s2_shared", 4, 128, 256, 128, 128, 3, 2, 1, "conv", False, torch.bfloat16),
    ("generic_odd_map_shared", 3, 72, 40, 45, 37, 3, 1, 1, "conv", False, torch.bfloat16),
    ("up2_per_sample", 3, 256, 256, 32, 32, 2, 1, 0, "up2", True, torch.bfloat16),
    ("up2_shared", 4, 128, 128, 32, 32, 2, 1, 0, "up2", False, torch.bfloat16),
    ("torgb_per_sample_split", 4, 512, 3, 128, 128, 1, 1, 0, "conv", True, torch.bfloat16),            # per-sample AND split
    ("small_map_big_output", 16, 1024, 1024, 16, 16, 3, 1, 1, "conv", False, torch.bfloat16),
    ("f32_shared", 4, 64, 96, 64, 64, 3, 1, 1, "conv", False, torch.float32),
    ("f32_generic_shared", 2, 40, 24, 33, 29, 3, 1, 1, "conv", False, torch.float32),
    ("f32_torgb_per_sample_split", 2, 256, 3, 128, 128, 1, 1, 0, "conv", True, torch.float32),
    ("thin_input", 4, 6, 128, 128, 128, 3, 1, 1, "conv", False, torch.bfloat16),
]


def _wgrad_reference(gy, x, k, stride, pad, kind, per_sample, o, i):
    """fp64 weight gradient, one contraction per tap (rocBLAS dgemm on the device: the CPU would take minutes on these sizes).
    conv: gw[o,i,kh,kw] = sum gy[b,o,y,x] * xpad[b,i,y*s+kh,x*s+kw];  up2: gw[o,i,dy,dx] = sum gy[b,o,2y+dy,2x+dx] * x[b,i,y,x]."""
    gy64, x64 = gy.double(), x.double()
    eq = "boyx,biyx->boi" if per_sample else "boyx,biyx->oi"
    taps = []
    if kind == "up2":
        for dy in range(2):
            for dx in range(2):
                taps.append(torch.einsum(eq, gy64[:, :, dy::2, dx::2], x64))
        kk = 2
    else:
        xp = torch.nn.functional.pad(x64, (pad, pad, pad, pad))
        oh, ow = gy64.shape[2:]
        for kh in range(k):
            for kw in range(k):
                taps.append(torch.einsum(eq, gy64, xp[:, :, kh:kh + stride * (oh - 1) + 1:stride,
                                                      kw:kw + stride * (ow - 1) + 1:stride]))
        kk = k
    out = torch.stack(taps, dim=-1)
    return out.reshape(*out.shape[:-1], kk, kk)


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_weight_gradient_kernels_are_deterministic(case):
    from multi_stylegan_amd import conv_ops
    _, b, i, o, h, w_, k, stride, pad, kind, per_sample, dtype = case
    torch.manual_seed(b * 1000 + i + o)
    x = conv_ops.to_compute_layout(torch.randn(b, i, h, w_, device=DEV), dtype)
    geo = conv_ops.Geometry(kind, k, k, stride, pad, (h, w_), per_sample)
    gy = conv_ops.to_compute_layout(torch.randn(b, o, *geo.y_hw, device=DEV), dtype)
    first = conv_ops._g_raw(gy, x, o, i, geo).clone()
    for _ in range(REPEATS):
        assert torch.equal(first, conv_ops._g_raw(gy, x, o, i, geo))
    # and the fixed-order sum is the right sum
    want = _wgrad_reference(gy, x, k, stride, pad, kind, per_sample, o, i)
    assert first.shape == want.shape
    assert rel_err(first, want) < (2e-5 if dtype == torch.float32 else 2e-3)


def test_weight_gradient_workspace_contract():
    """A split sum without (enough) workspace is refused, never silently accumulated; the query is a function of the shape."""
    from multi_stylegan_amd import _lib
    lib = _lib.lib()
    geom = (_lib.MSG_BF16, 4, 256, 256, 128, 128, 256, 256, 128, 128, 128, 3, 3, 1, 1, 0, 0, 1)
    need = lib.msg_conv2d_wgrad_workspace(*geom)
    assert need > 0 and need % (128 * 9 * 128) == 0 and need == lib.msg_conv2d_wgrad_workspace(*geom)
    x = torch.zeros(4, 256, 256, 128, dtype=torch.bfloat16, device=DEV)
    gw = torch.zeros(128, 9, 128, device=DEV)
    ws = torch.empty(need, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), x.data_ptr(), gw.data_ptr(), *geom, 0, 1.0)
    assert lib.msg_conv2d_wgrad(*args, None, 0, s) == -1                       # MSG_EINVAL
    assert lib.msg_conv2d_wgrad(*args, ws.data_ptr(), need - 1, s) == -1
    assert lib.msg_conv2d_wgrad(*args, ws.data_ptr(), need, s) == 0
    torch.cuda.synchronize()
    assert lib.msg_conv2d_wgrad_workspace(_lib.MSG_BF16, 1, 8, 8, 64, 64, 8, 8, 16, 16, 64, 3, 3, 1, 1, 0, 0, 1) == 0
    assert lib.msg_conv2d_wgrad_workspace(_lib.MSG_BF16, 1, 8, 8, 64, 64, 8, 8, 16, 16, 32, 3, 3, 1, 1, 0, 0, 1) == -1  # ldgw < I


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,channels_last,noise", [((16, 512, 64, 64), True, "batch"), ((4, 128, 256, 256), True, "shared"),
                                                       ((3, 20, 33, 29), False, "batch"), ((5, 7, 16, 16), True, None),
                                                       ((16, 512), False, None), ((2, 6, 40, 24), True, "batch")])
def test_activation_backward_sums_are_deterministic(shape, channels_last, noise, dtype):
    """grad_bias / grad_noise_weight of the fused activation: bit-identical over repeated launches, and equal to the sums."""
    from multi_stylegan_amd.op_static.fused_act import FusedLeakyReLUFunctionBackward
    torch.manual_seed(len(shape) * 100 + shape[1])
    g = torch.randn(shape, device=DEV).to(dtype)
    out = torch.randn(shape, device=DEV).to(dtype)
    if channels_last and len(shape) == 4:
        g, out = g.contiguous(memory_format=torch.channels_last), out.contiguous(memory_format=torch.channels_last)
    nz = None
    if noise is not None:
        nz = torch.randn((shape[0] if noise == "batch" else 1, 1, *shape[2:]), device=DEV)
    first = [t.clone() for t in FusedLeakyReLUFunctionBackward.apply(g, out, nz, True, 0.2, 1.5)]
    for _ in range(REPEATS):
        again = FusedLeakyReLUFunctionBackward.apply(g, out, nz, True, 0.2, 1.5)
        assert all(torch.equal(a, b) for a, b in zip(first, again))
    gx64 = g.double() * 1.5 * torch.where(out.double() > 0, 1.0, 0.2)
    dims = (0, 2, 3) if len(shape) == 4 else (0,)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(first[0].double(), gx64) < tol
    assert rel_err(first[1].double(), gx64.sum(dims)) < tol
    if nz is not None:
        assert rel_err(first[2].double(), (gx64 * nz.double()).sum().reshape(1)) < tol


def test_grouped_linear_latent_gradient_is_deterministic():
    from multi_stylegan_amd import conv_ops
    torch.manual_seed(4)
    g, b, l, n, k = 20, 16, 14, 512, 512
    slot = tuple(min(l - 1, j * l // g) for j in range(g))
    lat = torch.randn(b, l, k, device=DEV, requires_grad=True)
    ws = [torch.randn(n, k, device=DEV, requires_grad=True) for _ in range(g)]
    bs = [torch.randn(n, device=DEV, requires_grad=True) for _ in range(g)]
    gy = torch.randn(g, b, n, device=DEV)
    runs = []
    for _ in range(4):
        y = conv_ops._GroupedLinear.apply(lat, slot, 0.1, 1.0, *ws, *bs)
        runs.append(torch.autograd.grad(y, [lat, *ws], gy))
    for other in runs[1:]:
        assert all(torch.equal(a, c) for a, c in zip(runs[0], other))
    want = torch.zeros_like(lat)
    for j in range(g):
        want[:, slot[j]] += 0.1 * gy[j] @ ws[j].detach()
    assert rel_err(runs[0][0], want) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_iteration_is_bit_reproducible(golden, dtype):
    """Two trainers from the same state and the same draws, through a regularised iteration (D step, R1 with its double
    backward, G step, path length with its double backward, EMA): every parameter of G, D and the EMA copy, the Adam
    moments and the logged losses agree BIT FOR BIT.  With float atomics in the weight / bias gradients they did not."""
    import multi_stylegan_amd as m
    from test_hip_models import _golden_trainer
    from test_oracle_golden import load_train_draws
    results = []
    for _ in range(3):
        z, g, d, trainer = _golden_trainer(golden)
        g.compute_dtype = d.compute_dtype = trainer.generator_ema.compute_dtype = dtype
        real, draws = load_train_draws(z, 1, m.model_wrapper)           # golden iteration 16: both lazy regularisers fire
        trainer.iteration = 15
        trainer.train_iteration(real.to(DEV), draws.to(DEV))
        log = trainer.pop_logs()
        state = {f"G.{n}": p.detach().clone() for n, p in g.named_parameters()}
        state.update({f"D.{n}": p.detach().clone() for n, p in d.named_parameters()})
        state.update({f"E.{n}": p.detach().clone() for n, p in trainer.generator_ema.named_parameters()})
        names = {id(p): n for mod in (g, d) for n, p in mod.named_parameters()}
        for opt in (trainer.generator_optimizer, trainer.discriminator_optimizer):
            for p, st in opt.state.items():
                state[f"adam.m.{names[id(p)]}"], state[f"adam.v.{names[id(p)]}"] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
        results.append((state, log))
    (a, la) = results[0]
    assert "loss_discriminator_regularization" in la and "path_length" in la
    for b, lb in results[1:]:
        assert la == lb
        differing = [n for n in a if not torch.equal(a[n], b[n])]
        assert not differing, differing[:8]


def test_benchmark_configuration_is_bit_reproducible():
    """The same 17 training iterations twice at BASELINE configs[1] (256^2, batch 16, bf16 storage; the 16th iteration runs R1 and
    the path-length regulariser) from the same seeds: parameters, EMA copy and Adam moments bit-identical.  The tiny golden models
    of the test above never reach the large-tile kernels; this one runs every kernel of the benchmarked step -- the hand-scheduled
    K loops with their counted s_waitcnt, the LDS-DMA rings, the slab reductions -- where a piece that lands late or a race on an
    LDS stage shows up as a difference between two runs long before it shows up as a wrong loss (tools/determinism_soak.py: the
    same over more iterations)."""
    import hashlib
    import random
    import numpy as np
    import multi_stylegan_amd as m
    from multi_stylegan_amd.config import generator_config_for_resolution

    def run():
        torch.manual_seed(7); random.seed(7); np.random.seed(7)
        gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
        dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
        gen.compute_dtype = dis.compute_dtype = torch.bfloat16
        trainer = m.ModelWrapper(gen, dis, device=torch.device(DEV))
        trainer.generator_ema.compute_dtype = torch.bfloat16
        g = torch.Generator(device=DEV).manual_seed(11)
        for _ in range(17):
            trainer.train_iteration(torch.rand(16, 2, 3, 256, 256, device=DEV, generator=g))
        logs = trainer.pop_logs()
        assert "loss_discriminator_regularization" in logs and "path_length" in logs
        h = hashlib.sha256()
        tensors = [p for mod in (trainer.generator, trainer.discriminator, trainer.generator_ema) for p in mod.parameters()]
        for opt in (trainer.generator_optimizer, trainer.discriminator_optimizer):
            for st in opt.state.values():
                tensors += [v for v in st.values() if torch.is_tensor(v)]
        for t in tensors:
            h.update(t.detach().float().cpu().numpy().tobytes())
        return h.hexdigest(), len(tensors)

    (a, na), (b, nb) = run(), run()
    assert na == nb and na > 300
    assert a == b


def test_gradients_written_into_the_flat_store_equal_accumulated_gradients(golden):
    """dist.grad_destination: the weight / bias gradient kernels write straight into the parameters' slices of the flat
    gradient store (AccumulateGrad then adopts the tensor instead of launching `grad += incoming`).  Same training
    trajectory, bit for bit, as with every gradient going through autograd's accumulation add; and most parameters do take
    the direct route."""
    import multi_stylegan_amd as m
    from multi_stylegan_amd import dist as msg_dist
    from test_hip_models import _golden_trainer
    from test_oracle_golden import load_train_draws
    finals, taken = [], None
    for direct in (True, False):
        z, g, d, trainer = _golden_trainer(golden)
        trainer.generator_reducer.direct = trainer.discriminator_reducer.direct = direct
        for step, it in ((0, 1), (1, 16)):
            real, draws = load_train_draws(z, step, m.model_wrapper)
            trainer.iteration = it - 1
            trainer.train_iteration(real.to(DEV), draws.to(DEV))
        if direct:
            slots = [p.__dict__["_msg_grad_slot"] for p in d.parameters()]
            taken = sum(s.taken for s in slots) / len(slots)          # (state after the last discriminator backward)
        for red in (trainer.generator_reducer, trainer.discriminator_reducer):
            for b in red.buckets:
                for p, off in zip(b.params, b.offsets):               # every .grad is its view of the store again
                    assert p.grad is not None and p.grad.data_ptr() == b.flat.data_ptr() + 4 * off
        finals.append([p.detach().clone() for p in list(g.parameters()) + list(d.parameters())])
    assert taken is not None and taken > 0.6, taken
    assert all(torch.equal(a, b) for a, b in zip(*finals))


def test_ada_warp_backward_is_deterministic():
    """The augmentation warp's backward is a scatter; it accumulates in 64-bit fixed point, so repeated launches agree bit
    for bit (float atomics did not), and the result is the adjoint of the forward: <warp(x), g> == <x, warp^T(g)>."""
    from multi_stylegan_amd.adaptive_discriminator_augmentation import affine_warp
    torch.manual_seed(3)
    b, c, h, w = 8, 6, 96, 80
    x = torch.randn(b, c, h, w, device=DEV, requires_grad=True)
    u = torch.rand(b, device=DEV)
    p = torch.tensor(0.7, device=DEV)
    angle = torch.rand(b, device=DEV) * 360.0
    scale = torch.exp(torch.randn(b, 2, device=DEV) * 0.3)
    g = torch.randn(b, c, h, w, device=DEV) * 1e-4                     # (image gradients of mean-reduced losses are small)
    for padding in (0, 2):
        y = affine_warp(x, u, p, angle=angle, scale=scale, center=((w - 1) / 2, (h - 1) / 2), padding=padding,
                        align_corners=True)
        first, = torch.autograd.grad(y, x, g, retain_graph=True)
        for _ in range(REPEATS):
            again, = torch.autograd.grad(y, x, g, retain_graph=True)
            assert torch.equal(first, again)
        lhs, rhs = (y.detach().double() * g.double()).sum(), (x.detach().double() * first.double()).sum()
        assert abs(lhs - rhs) <= 1e-5 * abs(lhs), (padding, float(lhs), float(rhs))
        keep = (u > p)                                                 # images that were copied through: gradient == g exactly
        assert keep.any() and torch.equal(first[keep], g[keep])


@pytest.mark.parametrize("kind", ["conv_act_128", "modconv_act_512", "blur_act"])
def test_activation_backward_from_sign_bytes_is_bit_identical(kind, monkeypatch):
    """The forward kernels that can (row-sharing 3x3 conv, blur + activation) leave one sign byte per 8 output channels, and
    the activation's backward reads those instead of the stored output (msg_bias_act_backward_mask): every gradient bit for
    bit what the form that re-reads the output gives (MSG_ACT_MASK=0), and the masked kernel is the one that ran."""
    from multi_stylegan_amd import _lib, conv_ops
    from multi_stylegan_amd.op_static import fused_act, blur_bias_act
    torch.manual_seed(13)
    bf = torch.bfloat16

    def run(flag):
        monkeypatch.setattr(fused_act, "ACT_MASK", flag)
        torch.manual_seed(14)
        if kind == "conv_act_128":
            x = conv_ops.to_compute_layout(torch.randn(4, 128, 128, 128, device=DEV), bf).requires_grad_(True)
            w = (torch.randn(128, 128, 3, 3, device=DEV) / 34).requires_grad_(True)
            bias = torch.randn(128, device=DEV).requires_grad_(True)
            y = conv_ops.conv2d_bias_act(x, w, bias, padding=1, scale=math.sqrt(2))
            leaves = (x, w, bias)
        elif kind == "modconv_act_512":
            x = conv_ops.to_compute_layout(torch.randn(8, 512, 64, 64, device=DEV), bf).requires_grad_(True)
            w = torch.randn(1, 512, 512, 3, 3, device=DEV).requires_grad_(True)
            style = (1 + 0.1 * torch.randn(8, 512, device=DEV)).requires_grad_(True)
            bias = torch.randn(512, device=DEV).requires_grad_(True)
            noise = torch.randn(8, 1, 64, 64, device=DEV)
            nw = torch.full((1,), 0.3, device=DEV, requires_grad=True)
            y = conv_ops.modulated_conv2d_bias_act(x, w, style, True, bias, noise, nw, scale=math.sqrt(2))
            leaves = (x, w, style, bias, nw)
        else:
            x = conv_ops.to_compute_layout(torch.randn(2, 64, 33, 33, device=DEV), bf).requires_grad_(True)
            fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 16).to(DEV)
            bias = torch.randn(64, device=DEV).requires_grad_(True)
            noise = torch.randn(2, 1, 32, 32, device=DEV)
            nw = torch.full((1,), 0.3, device=DEV, requires_grad=True)
            y = blur_bias_act(x, fir, (1, 1), bias, noise, nw, scale=math.sqrt(2))
            leaves = (x, bias, nw)
        gy = torch.randn(y.shape, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
        _lib.kernel_clock.reset(enabled=True)
        grads = torch.autograd.grad(y, leaves, gy)
        torch.cuda.synchronize()
        keys = set(_lib.kernel_clock.summary())
        _lib.kernel_clock.reset(enabled=False)
        return y.detach(), grads, keys

    y1, g1, k1 = run(True)
    y0, g0, k0 = run(False)
    assert any(k.startswith("bias_act_bwd_mask/") for k in k1) and not any(k.startswith("bias_act_bwd_mask/") for k in k0)
    assert torch.equal(y1, y0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
