"""The MFMA contraction kernels (msg_conv2d_fprop / msg_conv2d_wgrad through multi_stylegan_amd.conv_ops) against
torch's CPU convolutions in fp64: forward, data gradient, weight gradient and one second-order term, for every
geometry of the path, shared and per-sample weights, ragged channel counts and tile tails.
f32 path = exact-fp32 MFMA: 1e-4 of max|ref| (BASELINE tolerance is 1e-3);  bf16 storage: 2e-2."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOLS = {torch.float32: 1e-4, torch.bfloat16: 2e-2}

CASES = [  # name, kind, B, I, O, H, W, k, stride, pad
    ("3x3_same", "conv", 2, 16, 24, 9, 11, 3, 1, 1),
    ("1x1", "conv", 2, 40, 8, 7, 5, 1, 1, 0),
    ("3x3_s2_p0_odd", "conv", 2, 8, 16, 17, 17, 3, 2, 0),
    ("3x3_same_ragged_ch", "conv", 1, 6, 3, 12, 12, 3, 1, 1),
    ("1x1_to_one", "conv", 2, 24, 1, 6, 6, 1, 1, 0),
    ("3x3_multi_tile", "conv", 1, 72, 136, 20, 20, 3, 1, 1),
    ("up2", "up2", 2, 16, 24, 5, 6, 2, 2, 0),
    ("up2_multi_tile", "up2", 1, 72, 40, 12, 12, 2, 2, 0),
    # power-of-two maps: the weight-gradient kernel's uniform-row buffer addressing, with the batch folded into K for
    # shared weights (maps smaller than / equal to / larger than one K-step, several K chunks, strided, up2)
    ("pow2_4x4_b5", "conv", 5, 16, 8, 4, 4, 3, 1, 1),
    ("pow2_8x8_b3", "conv", 3, 8, 16, 8, 8, 3, 1, 1),
    ("pow2_16x16_b3", "conv", 3, 16, 8, 16, 16, 3, 1, 1),
    ("pow2_64x64_b2", "conv", 2, 8, 8, 64, 64, 3, 1, 1),
    ("pow2_128w_1x1", "conv", 2, 16, 8, 4, 128, 1, 1, 0),
    ("pow2_s2_to_16", "conv", 2, 8, 8, 33, 33, 3, 2, 0),
    ("pow2_up2_8x8", "up2", 3, 8, 16, 8, 8, 2, 2, 0),
    # maps >= 64 wide: the three-taps-per-workgroup weight-gradient kernel (conv_wgrad_row3.hip), several segments per
    # row, ragged channel tiles, several K chunks
    ("row3_128w_ragged", "conv", 3, 72, 136, 5, 128, 3, 1, 1),
    ("row3_64w_tall", "conv", 2, 24, 40, 70, 64, 3, 1, 1),
    ("row3_512w", "conv", 2, 16, 24, 3, 512, 3, 1, 1),
    ("row3_32w", "conv", 3, 24, 40, 32, 32, 3, 1, 1),                            # two image rows per K-step
    # maps a column or two short of a power of two (the discriminator's stride-2 outputs: 127, 63, 31, 15): the weight-gradient
    # kernel's uniform rows on rows padded to that power of two (folded batch: 63 / 127 only), the others as they are
    ("s2_to_63", "conv", 2, 8, 16, 127, 127, 3, 2, 0),
    ("s2_to_31_b3", "conv", 3, 16, 8, 63, 63, 3, 2, 0),
    ("s2_to_15", "conv", 2, 8, 8, 31, 31, 3, 2, 0),
    ("same_63w", "conv", 2, 16, 24, 5, 63, 3, 1, 1),
    ("1x1_127w", "conv", 2, 24, 8, 3, 127, 1, 1, 0),
    # K loops of one, two and four K-steps: the three-stage ring's prologue issues three steps whatever the length
    ("row3_one_step", "conv", 1, 8, 16, 1, 64, 3, 1, 1),
    ("row3_two_steps", "conv", 1, 16, 8, 1, 128, 3, 1, 1),
    ("row3_four_steps", "conv", 2, 8, 8, 2, 64, 3, 1, 1),
    ("row3_32w_ragged", "conv", 2, 136, 72, 6, 32, 3, 1, 1),
]


def _reference(kind, x, w, stride, pad, per_sample):
    """fp64 CPU reference; w is [O,I,kh,kw] or [B,O,I,kh,kw]."""
    def one(xb, wb):
        if kind == "up2":
            return F.conv_transpose2d(xb, wb.transpose(0, 1), stride=2)
        return F.conv2d(xb, wb, stride=stride, padding=pad)
    if per_sample:
        return torch.cat([one(x[b:b + 1], w[b]) for b in range(x.shape[0])])
    return one(x, w)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_primitives(case, per_sample, dtype):
    from multi_stylegan_amd import conv_ops
    name, kind, b, i, o, h, w_, k, stride, pad = case
    tol = TOLS[dtype]
    g = torch.Generator().manual_seed(len(name) * 7 + b)
    x = torch.randn(b, i, h, w_, generator=g).to(dtype).double()
    wshape = (b, o, i, k, k) if per_sample else (o, i, k, k)
    w = (torch.randn(*wshape, generator=g) / math.sqrt(i * k * k)).to(dtype).double()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = _reference(kind, xr, wr, stride, pad, per_sample)
    gy = torch.randn(yr.shape, generator=g).to(dtype).double()
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy, create_graph=True)
    # second order: d/dw of <gx, v>  and d/dx of <gw, u>
    v = torch.randn(x.shape, generator=g).to(dtype).double()
    u = torch.randn(w.shape, generator=g).double()
    ggw_r, = torch.autograd.grad(gxr, wr, v, retain_graph=True)
    ggx_r, = torch.autograd.grad(gwr, xr, u, retain_graph=True)

    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV, torch.float32).requires_grad_(True)
    geo = conv_ops.Geometry(kind, k, k, stride if kind == "conv" else 1, pad, (h, w_), per_sample)
    y = conv_ops._ConvF.apply(xd, wd, None, geo)
    assert y.shape == yr.shape and y.dtype == dtype
    gyd = gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    gx, gw = torch.autograd.grad(y, (xd, wd), gyd, create_graph=True)
    ggw, = torch.autograd.grad(gx, wd, v.to(DEV, dtype).contiguous(memory_format=torch.channels_last),
                               retain_graph=True)
    ggx, = torch.autograd.grad(gw, xd, u.to(DEV, torch.float32), retain_graph=True)
    assert rel_err(y.float(), yr) < tol, "forward"
    assert rel_err(gx.float(), gxr) < tol, "data gradient"
    assert rel_err(gw, gwr) < tol, "weight gradient"
    assert rel_err(ggw, ggw_r) < tol, "d(dgrad)/dw"
    assert rel_err(ggx.float(), ggx_r) < tol, "d(wgrad)/dx"


@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_primitives_split_bf16_products(case, per_sample, mode="split_bf16x3", tol=2e-6):
    """The fp32-storage contractions with MSG_F32_SPLIT (conv_ops.fp32_contraction("split_bf16x3")): every product as six bf16
    MFMA products on (hi, mid, lo) splits, fp32 accumulation.  Forward, data gradient, weight gradient and the two
    second-order contractions on operands with FULL fp32 mantissas against the fp64 reference: at the exact-fp32 kernel's own
    error (2e-6 of max|ref|, or twice the exact kernel's; a plain bf16 product of these operands is off by 4e-3), and
    measurably not the exact kernel (the mode really ran)."""
    from multi_stylegan_amd import conv_ops
    name, kind, b, i, o, h, w_, k, stride, pad = case
    g = torch.Generator().manual_seed(len(name) * 11 + b)
    x = torch.randn(b, i, h, w_, generator=g).double()
    wshape = (b, o, i, k, k) if per_sample else (o, i, k, k)
    w = (torch.randn(*wshape, generator=g) / math.sqrt(i * k * k)).double()
    x, w = x.float().double(), w.float().double()                     # fp32-representable, 24-bit mantissas
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = _reference(kind, xr, wr, stride, pad, per_sample)
    gy = torch.randn(yr.shape, generator=g).float().double()
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy, create_graph=True)
    v = torch.randn(x.shape, generator=g).float().double()
    u = torch.randn(w.shape, generator=g).float().double()
    ggw_r, = torch.autograd.grad(gxr, wr, v, retain_graph=True)
    ggx_r, = torch.autograd.grad(gwr, xr, u, retain_graph=True)
    cl = lambda t: t.to(DEV, torch.float32).contiguous(memory_format=torch.channels_last)

    def run(forward_mode=None):
        """forward_mode: run ONLY the forward inside ``with fp32_contraction(forward_mode)`` -- the layer's geometry captures the
        mode there, and the backward / double backward launched after the block must multiply the same way."""
        xd, wd = cl(x).requires_grad_(True), w.to(DEV, torch.float32).requires_grad_(True)
        if forward_mode is None:
            geo = conv_ops.Geometry(kind, k, k, stride if kind == "conv" else 1, pad, (h, w_), per_sample)
            y = conv_ops._ConvF.apply(xd, wd, None, geo)
        else:
            with conv_ops.fp32_contraction(forward_mode):
                geo = conv_ops.Geometry(kind, k, k, stride if kind == "conv" else 1, pad, (h, w_), per_sample)
                y = conv_ops._ConvF.apply(xd, wd, None, geo)
        gx, gw = torch.autograd.grad(y, (xd, wd), cl(gy) if y.shape[1] > 1 else gy.to(DEV, torch.float32), create_graph=True)
        ggw, = torch.autograd.grad(gx, wd, cl(v), retain_graph=True)
        ggx, = torch.autograd.grad(gw, xd, u.to(DEV, torch.float32), retain_graph=True)
        return y.detach(), gx.detach(), gw.detach(), ggw.detach(), ggx.detach()

    exact = run()
    with conv_ops.fp32_contraction(mode):
        split = run()
    assert conv_ops.FP32_CONTRACTION == "exact"
    # (advisor, round 4: the mode used to be read at LAUNCH time, so a backward after the block silently ran exact kernels)
    late = run(forward_mode=mode)
    assert all(torch.equal(a, b_) for a, b_ in zip(late, split)), "backward after the block did not keep the forward's mode"
    for nm, got, ex, ref in zip(("forward", "data gradient", "weight gradient", "d(dgrad)/dw", "d(wgrad)/dx"), split, exact,
                                (yr, gxr, gwr, ggw_r, ggx_r)):
        # split_bf16x3 (all 24 mantissa bits, six products) sits at the exact kernel's own error, fp32 rounding of the sums
        assert rel_err(got.float(), ref) < max(tol, 2.0 * rel_err(ex.float(), ref)), (nm, rel_err(got.float(), ref), rel_err(ex.float(), ref))
    assert any(not torch.equal(a, e) for a, e in zip(split, exact)), "the split products did not run"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_bias_and_linear(dtype):
    from multi_stylegan_amd import conv_ops
    tol = TOLS[dtype]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 9, 9, generator=g).to(dtype).double()
    w = (torch.randn(24, 16, 3, 3, generator=g) / 12).double()
    bias = torch.randn(24, generator=g).double()
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, bias)]
    yr = F.conv2d(xr, wr, br, stride=2)
    gy = torch.randn(yr.shape, generator=g).to(dtype).double()
    gr = torch.autograd.grad(yr, (xr, wr, br), gy)
    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(DEV, torch.float32).requires_grad_(True), bias.to(DEV, torch.float32).requires_grad_(True)
    y = conv_ops.conv2d(xd, wd, bd, stride=(2, 2), padding=(0, 0))
    gd = torch.autograd.grad(y, (xd, wd, bd), gy.to(DEV, dtype))
    assert rel_err(y.float(), yr) < tol
    for a, r in zip(gd, gr):
        assert rel_err(a.float(), r) < tol * (4 if dtype == torch.bfloat16 else 1)
    # linear: [B,I] x [O,I]^T + b with B not a multiple of anything
    xl = torch.randn(5, 48, generator=g).to(dtype).double().requires_grad_(True)
    wl = (torch.randn(20, 48, generator=g) / 7).double().requires_grad_(True)
    bl = torch.randn(20, generator=g).double().requires_grad_(True)
    yl = F.linear(xl, wl, bl)
    gyl = torch.randn(yl.shape, generator=g).to(dtype).double()
    grl = torch.autograd.grad(yl, (xl, wl, bl), gyl)
    xdl = xl.detach().to(DEV, dtype).requires_grad_(True)
    wdl, bdl = wl.detach().to(DEV, torch.float32).requires_grad_(True), bl.detach().to(DEV, torch.float32).requires_grad_(True)
    ydl = conv_ops.linear(xdl, wdl, bdl)
    gdl = torch.autograd.grad(ydl, (xdl, wdl, bdl), gyl.to(DEV, dtype))
    assert rel_err(ydl.float(), yl) < tol
    for a, r in zip(gdl, grl):
        assert rel_err(a.float(), r) < tol * (4 if dtype == torch.bfloat16 else 1)


def test_conv_full_size_properties():
    """BASELINE-size 3x3 (512 -> 512 @ 64^2, B=2, bf16) -- linearity in x and <F(x), g> == <x, D(g)> == <w, G(g, x)>."""
    from multi_stylegan_amd import conv_ops
    torch.manual_seed(2)
    x = torch.randn(2, 512, 64, 64, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn_like(x)
    w = (torch.randn(512, 512, 3, 3, device=DEV) / math.sqrt(512 * 9)).requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    y = conv_ops.conv2d(xg, w, padding=1)
    y2 = conv_ops.conv2d(x2, w.detach(), padding=1)
    ys = conv_ops.conv2d(x + x2, w.detach(), padding=1)
    assert rel_err(ys.float(), y.float() + y2.float()) < 2e-2
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (xg, w), gy)
    a = (y.double() * gy.double()).sum()
    b = (x.double() * gx.double()).sum()
    c = (w.double() * gw.double()).sum()
    assert abs(a - b) / abs(a) < 2e-2 and abs(a - c) / abs(a) < 2e-2
    # against the library conv on the same bf16 data
    ref = torch.nn.functional.conv2d(x.float(), w.detach(), padding=1)
    assert rel_err(y.float(), ref) < 2e-2


def _random_cases(n=36, seed=20260101):
    """Random geometries over everything the addressing paths branch on: power-of-two / odd maps, maps smaller and larger
    than a K-step, few / ragged / multi-tile channel counts, strides, kernel sizes, shared and per-sample weights."""
    import random
    rng = random.Random(seed)
    cases = []
    while len(cases) < n:
        kind = rng.choice(["conv", "conv", "conv", "up2"])
        b = rng.choice([1, 2, 3, 5])
        i = rng.choice([3, 6, 8, 16, 40, 72, 136])
        o = rng.choice([1, 3, 8, 24, 72, 136, 264])
        h = rng.choice([4, 7, 8, 15, 16, 31, 32, 64])
        w_ = rng.choice([4, 5, 8, 16, 17, 32, 64, 128]) if rng.random() < 0.5 else h
        if kind == "up2":
            k, stride, pad = 2, 2, 0
        else:
            k = rng.choice([1, 3, 3, 3])
            stride = rng.choice([1, 1, 1, 2])
            pad = rng.choice([0, k // 2])
            if (h + 2 * pad - k) // stride + 1 <= 0 or (w_ + 2 * pad - k) // stride + 1 <= 0:
                continue
        if b * max(i, o) * h * w_ > 3_000_000:          # keep the fp64 CPU reference quick
            continue
        cases.append((f"r{len(cases)}_{kind}_b{b}_i{i}_o{o}_{h}x{w_}_k{k}s{stride}p{pad}", kind, b, i, o, h, w_, k, stride, pad))
    return cases


RANDOM_CASES = _random_cases()


@pytest.mark.parametrize("case", RANDOM_CASES, ids=[c[0] for c in RANDOM_CASES])
def test_conv_primitives_random_geometries(case):
    """F / D / G on randomly drawn geometries (bf16 storage, shared and per-sample weights) against the fp64 reference."""
    from multi_stylegan_amd import conv_ops
    name, kind, b, i, o, h, w_, k, stride, pad = case
    dtype, tol = torch.bfloat16, TOLS[torch.bfloat16]
    g = torch.Generator().manual_seed(sum(ord(ch) for ch in name) * 7 + len(name))
    for per_sample in (False, True):
        x = torch.randn(b, i, h, w_, generator=g).to(dtype).double()
        wshape = (b, o, i, k, k) if per_sample else (o, i, k, k)
        w = (torch.randn(*wshape, generator=g) / math.sqrt(i * k * k)).to(dtype).double()
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = _reference(kind, xr, wr, stride, pad, per_sample)
        gy = torch.randn(yr.shape, generator=g).to(dtype).double()
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy)
        xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.to(DEV, torch.float32).requires_grad_(True)
        geo = conv_ops.Geometry(kind, k, k, stride if kind == "conv" else 1, pad, (h, w_), per_sample)
        y = conv_ops._ConvF.apply(xd, wd, None, geo)
        gyd = gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last) if y.shape[1] > 1 else gy.to(DEV, dtype)
        gx, gw = torch.autograd.grad(y, (xd, wd), gyd)
        assert y.shape == yr.shape
        assert rel_err(y.float(), yr) < tol, ("forward", per_sample)
        assert rel_err(gx.float(), gxr) < tol, ("data gradient", per_sample)
        assert rel_err(gw, gwr) < tol, ("weight gradient", per_sample)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 24, 12, 12, 3), (1, 72, 136, 20, 20, 3), (3, 40, 9, 8, 8, 1),
                                   (2, 64, 256, 64, 64, 3)], ids=["small", "multi_tile", "ragged_1x1", "pp_tile"])
def test_conv_with_fused_activation_is_bit_identical(shape, dtype):
    """msg_conv2d_fprop_act == msg_conv2d_fprop followed by msg_fused_bias_act, bit for bit (the activation is applied
    to the rounded conv result in the epilogue), and so are first- and second-order gradients."""
    from multi_stylegan_amd import conv_ops
    from multi_stylegan_amd.op_static import fused_leaky_relu
    b, i, o, h, w_, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(b, i, h, w_, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(o, i, k, k, generator=g) / math.sqrt(i * k * k)).to(DEV)
    bias = torch.randn(o, generator=g).to(DEV)
    gy = torch.randn(b, o, h, w_, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    v = torch.randn(b, i, h, w_, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    outs = []
    for fused in (True, False):
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        if fused:
            y = conv_ops.conv2d_bias_act(xs, ws, bs, stride=1, padding=k // 2, wscale=0.7, negative_slope=0.2, scale=1.4)
        else:
            y = fused_leaky_relu(conv_ops.conv2d(xs, ws, None, stride=1, padding=k // 2, wscale=0.7), bs, 0.2, 1.4)
        gx, gw, gb = torch.autograd.grad(y, (xs, ws, bs), gy, create_graph=True)
        ggw, = torch.autograd.grad(gx, ws, v, retain_graph=True)        # R1-type second-order term
        outs.append((y, gx, gw, gb, ggw))
    for name, a, r in zip(("y", "gx", "gw", "gb", "ggw"), *outs):
        if name in ("gw", "gb", "ggw"):                                 # float atomics: order-dependent rounding
            assert rel_err(a.float(), r.float()) < 1e-5, name
        else:
            assert torch.equal(a, r), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("noise_batch", [1, 3])
def test_modulated_conv_with_fused_activation_is_bit_identical(noise_batch, dtype):
    from multi_stylegan_amd import conv_ops
    from multi_stylegan_amd.op_static import fused_bias_noise_leaky_relu
    b, i, o, h = 3, 24, 40, 16
    g = torch.Generator().manual_seed(77 + noise_batch)
    x = torch.randn(b, i, h, h, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, o, i, 3, 3, generator=g).to(DEV)
    style = (torch.randn(b, i, generator=g) * 0.3 + 1).to(DEV)
    bias, nw = torch.randn(o, generator=g).to(DEV), torch.tensor([0.37], device=DEV)
    noise = torch.randn(noise_batch, 1, h, h, generator=g).to(DEV)
    gy = torch.randn(b, o, h, h, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (x, w, style, bias, nw)]
        xs, ws, ss, bs, ns = leaves
        if fused:
            y = conv_ops.modulated_conv2d_bias_act(xs, ws, ss, True, bs, noise, ns, 0.2, 1.0)
        else:
            y = fused_bias_noise_leaky_relu(conv_ops.modulated_conv2d(xs, ws, ss, True, False), bs, noise, ns, 0.2, 1.0)
        first = torch.autograd.grad(y, leaves, gy, retain_graph=True)                     # fused first-order path
        gs, = torch.autograd.grad(y, ss, gy, create_graph=True)                           # path-length-type term
        second = torch.autograd.grad(gs.square().sum(), [ws, ss])
        res.append((y, first, second))
    assert torch.equal(res[0][0], res[1][0])
    for a, r in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert rel_err(a.float(), r.float()) < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(24, 16, 3), (136, 72, 3), (8, 40, 1), (40, 24, 2), (3, 512, 1), (512, 6, 3)],
                         ids=["small", "multi_tile", "1x1", "2x2", "to_rgb", "from_rgb"])
def test_fused_weight_relayout_matches_torch_relayout(shape, dtype):
    """csrc/relayout.hip (one kernel: forward image, data-gradient image, sum of squared taps) against the torch
    transpose-copy functions it replaces, bit for bit, for the plain and the transposed-conv (up2) layouts."""
    from multi_stylegan_amd import conv_ops
    o, i, k = shape
    torch.manual_seed(o + i)
    w = torch.nn.Parameter(torch.randn(o, i, k, k, device=DEV))
    for kind in (("conv", "up2") if k == 2 else ("conv",)):
        img = conv_ops._param_images(w, dtype, 0.37, kind)
        assert img is not None
        f_ref, ck = conv_ops._relay_fwd_kind(w.detach(), dtype, kind)
        d_ref, ok = conv_ops._relay_dgrad(w.detach(), dtype, flip=kind != "up2")
        # reference path scales AFTER rounding to the storage type; the fused kernel scales in fp32 and rounds once
        tol = 0 if dtype == torch.float32 else 1e-2
        assert img["f"][1] == ck and img["d"][1] == ok
        assert img["f"][0].shape == f_ref.shape and img["d"][0].shape == d_ref.shape
        assert rel_err(img["f"][0].float(), (f_ref.float() * 0.37)) <= tol + 1e-7
        assert rel_err(img["d"][0].float(), (d_ref.float() * 0.37)) <= tol + 1e-7
        assert (img["f"][0][..., i:] == 0).all() and (img["d"][0][..., o:] == 0).all()          # padding is zero
    mod = conv_ops._param_images(w, torch.float32, 1.0, "conv", modulation=True)
    w3 = w.detach().reshape(o, i, k * k)
    assert torch.equal(mod["f"][0], w3.transpose(1, 2).contiguous())
    assert torch.equal(mod["d"][0], w3.flip(-1).permute(1, 2, 0).contiguous())
    assert rel_err(mod["wsq"], w3.square().sum(dim=2)) < 1e-6
    # cache: same object until the weight generation changes
    assert conv_ops._param_images(w, dtype, 0.37, "conv") is conv_ops._param_images(w, dtype, 0.37, "conv")
    conv_ops.invalidate_weight_cache()
    assert conv_ops._param_images(w, torch.float32, 1.0, "conv", modulation=True) is not mod


@pytest.mark.parametrize("m,n,k", [(16, 512, 512), (5, 7, 70), (33, 20, 1100), (1, 1, 3), (64, 129, 257)])
def test_few_row_linear_family(m, n, k):
    """csrc/linear.hip: forward, both gradients, the fused bias gradient and the second-order terms R1 / path length
    need, against float64 autograd of F.linear(x, w * scale, b)."""
    from multi_stylegan_amd import conv_ops
    g = torch.Generator().manual_seed(m * 1000 + n)
    scale = 0.37
    x = torch.randn(m, k, generator=g).double().requires_grad_(True)
    w = torch.randn(n, k, generator=g).double().requires_grad_(True)
    b = torch.randn(n, generator=g).double().requires_grad_(True)
    y = F.linear(x, w * scale, b * 0.61)
    gy = torch.randn(m, n, generator=g).double()
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy, create_graph=True)
    u, v = torch.randn(m, k, generator=g).double(), torch.randn(n, k, generator=g).double()
    ggw_r, = torch.autograd.grad(gx, w, u, retain_graph=True)             # d<gx,u>/dw
    ggx_r, = torch.autograd.grad(gw, x, v, retain_graph=True)             # d<gw,v>/dx
    xd, wd, bd = [t.detach().to(DEV, torch.float32).requires_grad_(True) for t in (x, w, b)]
    yd = conv_ops.linear(xd, wd, bd, wscale=scale, bias_scale=0.61)
    assert yd.dtype == torch.float32 and yd.shape == (m, n)
    gyd = gy.to(DEV, torch.float32)
    gxd, gwd, gbd = torch.autograd.grad(yd, (xd, wd, bd), gyd, create_graph=True)
    ggw, = torch.autograd.grad(gxd, wd, u.to(DEV, torch.float32), retain_graph=True)
    ggx, = torch.autograd.grad(gwd, xd, v.to(DEV, torch.float32), retain_graph=True)
    tol = 2e-5
    for name, a, r in [("y", yd, y), ("gx", gxd, gx), ("gw", gwd, gw), ("gb", gbd, gb), ("ggw", ggw, ggw_r),
                       ("ggx", ggx, ggx_r)]:
        assert rel_err(a, r) < tol, name
    # first-order step (no graph): weight and bias gradient come from ONE launch
    yd2 = conv_ops.linear(xd, wd, bd, wscale=scale, bias_scale=0.61)
    g1 = torch.autograd.grad(yd2, (xd, wd, bd), gyd)
    for a, r in zip(g1, (gx, gw, gb)):
        assert rel_err(a, r) < tol
    # empty batch
    e = conv_ops.linear(torch.zeros(0, k, device=DEV, requires_grad=True), wd, bd)
    assert e.shape == (0, n)
    gwe, = torch.autograd.grad(e.sum(), wd)
    assert gwe.abs().max().item() == 0.0


@pytest.mark.parametrize("batch", [3, 21])          # 21: more than the 16 samples one backward launch takes (chunked)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["conv3x3_demod", "up2x2_demod", "torgb1x1_nodemod"])
def test_fused_modulated_conv_matches_composite(kind, dtype, batch):
    """csrc/modulate.hip path (demod reduction, per-sample weights in kernel layout, fused backward) against the
    composite torch-op formulation AND against the CPU oracle, forward and all three gradients."""
    from multi_stylegan_amd import conv_ops
    from oracle import ops as oo
    tol = TOLS[dtype] * (3 if dtype == torch.float32 else 1)
    k = {"conv3x3_demod": 3, "up2x2_demod": 2, "torgb1x1_nodemod": 1}[kind]
    demod, up = kind != "torgb1x1_nodemod", kind == "up2x2_demod"
    b, i, o, h = batch, 40, (3 if k == 1 else 24), 10
    g = torch.Generator().manual_seed(k)
    x = torch.randn(b, i, h, h, generator=g).to(dtype).float()
    w = torch.randn(1, o, i, k, k, generator=g)
    s = torch.randn(b, i, generator=g) + 1.0
    xr, wr, sr = [t.clone().double().requires_grad_(True) for t in (x, w, s)]
    if up:
        import math
        scale = math.sqrt(2.0) / math.sqrt(i * k * k)
        wm = (scale * wr) * sr.reshape(b, 1, i, 1, 1)
        wm = wm * torch.rsqrt(wm.square().sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
        yr = torch.cat([F.conv_transpose2d(xr[n:n + 1], wm[n].transpose(0, 1), stride=2) for n in range(b)])
    else:
        yr = oo.modulated_conv2d(xr, wr, sr, demodulate=demod, upsample=False)
    gy = torch.randn(yr.shape, generator=g).to(dtype).double()
    gr = torch.autograd.grad(yr, (xr, wr, sr), gy)
    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, sd = w.to(DEV).requires_grad_(True), s.to(DEV).requires_grad_(True)
    gyd = gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    y = conv_ops.modulated_conv2d(xd, wd, sd, demod, up)
    calls = []
    orig = conv_ops._modconv_backward
    conv_ops._modconv_backward = lambda *a, **k: (calls.append(a[0].shape[0]), orig(*a, **k))[1]
    try:
        gd = torch.autograd.grad(y, (xd, wd, sd), gyd)
    finally:
        conv_ops._modconv_backward = orig
    assert calls == ([3] if batch == 3 else [16, 5]), "the fused backward must run, in chunks of at most 16 samples"
    yc = conv_ops._modulated_composite(xd, wd, sd, demod, up)
    gc = torch.autograd.grad(yc, (xd, wd, sd), gyd)
    assert rel_err(y.float(), yr) < tol and rel_err(y.float(), yc.float()) < tol
    for a, c, r, name in zip(gd, gc, gr, ("gx", "gw", "gs")):
        assert rel_err(a.float(), r) < tol, name
        assert rel_err(a.float(), c.float()) < tol, name + " vs composite"


def _random_mod_cases(n=14, seed=77):
    import random
    rng = random.Random(seed)
    out = []
    for idx in range(n):
        up = rng.random() < 0.3
        k = 2 if up else rng.choice([1, 3])
        out.append((f"m{idx}", rng.choice([1, 2, 4]), rng.choice([8, 24, 72, 136, 520]), rng.choice([3, 8, 40, 136]),
                    rng.choice([4, 8, 13, 16, 32]), k, up, rng.random() < 0.7, rng.random() < 0.5))
    return out


RANDOM_MOD_CASES = _random_mod_cases()


@pytest.mark.parametrize("case", RANDOM_MOD_CASES, ids=[c[0] for c in RANDOM_MOD_CASES])
def test_modulated_conv_random_geometries(case):
    """Fused modulated conv (one-launch weight set, per-sample contraction, fused backward; optionally the activation in
    the epilogue) against the composite torch-op formulation on random channel counts / taps / map sizes, incl. channel
    counts beyond the fused backward's limits (I > 512 falls back to the composite inside the op)."""
    from multi_stylegan_amd import conv_ops
    from multi_stylegan_amd.op_static import fused_bias_noise_leaky_relu
    name, b, i, o, h, k, up, demod, with_act = case
    if up and o % 8:
        o = 8 * ((o + 7) // 8)
    dtype, tol = torch.bfloat16, TOLS[torch.bfloat16] * 1.5
    g = torch.Generator().manual_seed(sum(ord(c) for c in name) + i + o)
    x = torch.randn(b, i, h, h, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    w = torch.randn(1, o, i, k, k, generator=g).to(DEV)
    st = (torch.randn(b, i, generator=g) * 0.3 + 1).to(DEV)
    oh = 2 * h if up else h
    gy = torch.randn(b, o, oh, oh, generator=g).to(DEV, dtype)
    gy = gy.contiguous(memory_format=torch.channels_last) if o > 1 else gy
    bias, nw = torch.randn(o, generator=g).to(DEV), torch.tensor([0.3], device=DEV)
    noise = torch.randn(b, 1, oh, oh, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (x, w, st)]
        xs, ws, ss = leaves
        if fused:
            if with_act and not up:
                y = conv_ops.modulated_conv2d_bias_act(xs, ws, ss, demod, bias, noise, nw, 0.2, 1.0)
            else:
                y = conv_ops.modulated_conv2d(xs, ws, ss, demod, up)
        else:
            y = conv_ops._modulated_composite(xs, ws, ss, demod, up)
            if with_act and not up:
                y = fused_bias_noise_leaky_relu(y, bias, noise, nw, 0.2, 1.0)
        res.append((y, torch.autograd.grad(y, leaves, gy)))
    assert rel_err(res[0][0].float(), res[1][0].float()) < tol
    for a, r, nm in zip(res[0][1], res[1][1], ("gx", "gw", "gs")):
        assert (a.float() - r.float()).norm() / (r.float().norm() + 1e-12) < tol, nm


PP_CASES = [  # shapes that take the 256x256 ping-pong kernel: name, kind, B, I, O, H, W, k, stride, pad, per_sample
    ("pp_3x3_ragged_m", "conv", 5, 64, 256, 15, 15, 3, 1, 1, True),            # per-sample M = 225 (< tile), many z
    ("pp_3x3_shared", "conv", 4, 72, 384, 40, 40, 3, 1, 1, False),              # N tail (384), ragged K (72)
    ("pp_1x1_shared", "conv", 8, 256, 256, 48, 48, 1, 1, 0, False),
    ("pp_up2", "up2", 3, 64, 128, 36, 36, 2, 2, 0, True),                        # 4*O = 512 columns, pixel shuffle
    ("pp_s2_dgrad", "conv", 4, 256, 256, 65, 65, 3, 2, 0, False),               # its data gradient uses in_up = 2
    # 3x3 'same' convs on maps 64 / 128 / 256 wide: conv_fprop_row3.hip (activation tile shared by the horizontal taps);
    # several image-row segments per tile, ragged K, N tail, shared and per-sample weights, many / few rows
    ("row3_64w", "conv", 2, 72, 256, 40, 64, 3, 1, 1, False),
    ("row3_128w_ps", "conv", 2, 64, 512, 10, 128, 3, 1, 1, True),
    ("row3_256w", "conv", 1, 128, 288, 5, 256, 3, 1, 1, False),
    ("row3_64w_ps_ragged_n", "conv", 3, 136, 264, 64, 64, 3, 1, 1, True),
    ("row3_512w_ps", "conv", 8, 64, 256, 16, 512, 3, 1, 1, True),                # tiles start mid-row: live left / right neighbours
    # exactly 256 wide, > 128 channels, N a multiple of 256: one whole image row per 256 x 256 tile (the benchmark's largest
    # launches), both halo rows outside the image; top / bottom image rows, 3 and 4 channel chunks
    ("row3_256w_ps_192", "conv", 2, 192, 512, 60, 256, 3, 1, 1, True),
    ("row3_256w_shared_256", "conv", 4, 256, 256, 64, 256, 3, 1, 1, False),
    # the 128 x 128 variant (two workgroups per CU) for 128 / 384 output channels
    ("row3_narrow_128", "conv", 4, 128, 128, 64, 128, 3, 1, 1, False),
    ("row3_narrow_384", "conv", 8, 64, 384, 64, 64, 3, 1, 1, False),
    ("row3_narrow_ps_136", "conv", 8, 72, 136, 128, 256, 3, 1, 1, True),
    ("row3_32w_768", "conv", 8, 64, 768, 32, 32, 3, 1, 1, False),                # four image rows per 128-pixel tile
    ("row3_32w_ps_264", "conv", 16, 136, 264, 32, 32, 3, 1, 1, True),
]


@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_pingpong_kernel_shapes(case):
    """bf16 forward + data gradient on shapes large enough for conv_fprop_pp.hip, against fp64 CPU convolutions."""
    from multi_stylegan_amd import conv_ops
    name, kind, b, i, o, h, w_, k, stride, pad, per_sample = case
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(b, i, h, w_, generator=g).bfloat16().double()
    wshape = (b, o, i, k, k) if per_sample else (o, i, k, k)
    w = (torch.randn(*wshape, generator=g) / math.sqrt(i * k * k)).bfloat16().double()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = _reference(kind, xr, wr, stride, pad, per_sample)
    gy = torch.randn(yr.shape, generator=g).bfloat16().double()
    gxr, = torch.autograd.grad(yr, xr, gy)
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV, torch.float32)
    geo = conv_ops.Geometry(kind, k, k, stride if kind == "conv" else 1, pad, (h, w_), per_sample)
    y = conv_ops._ConvF.apply(xd, wd, None, geo)
    gx, = torch.autograd.grad(y, xd, gy.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert rel_err(y.float(), yr) < 2e-2, "forward"
    assert rel_err(gx.float(), gxr) < 2e-2, "data gradient"


@pytest.mark.parametrize("b", [31, 33], ids=["just_below_2GiB", "above_2GiB"])
@pytest.mark.parametrize("k,o,with_wgrad", [(3, 128, True), (1, 256, False)], ids=["3x3_512to128", "1x1_512to256"])
def test_conv_above_two_gib_of_activations(k, o, with_wgrad, b):
    """Shared-weight convolution over a batch whose activations exceed 2 GiB (33 x 512 x 256 x 256 bf16): the kernels that address
    their operands through 31-bit buffer offsets (conv_fprop_row3 / conv_fprop_pp / conv_wgrad_row3 / conv_wgrad's uniform rows)
    must hand the launch to the 64-bit-pointer kernels (their eligibility tests), and the result must agree with the same
    convolution run on slices of the batch that ARE eligible -- forward and data gradient on the first and the LAST samples (the
    last one starts beyond 2 GiB), the weight gradient as the sum over three thirds of the batch."""
    from multi_stylegan_amd import conv_ops
    i, r = 512, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(b, i, r, r, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)
    # (31 samples: 2^31 - 2^26 bytes, the descriptor-addressed kernels still take it with offsets up to the last bit of their
    #  range; 33 samples: past it)
    assert (x.numel() * 2 > (1 << 31)) == (b == 33)
    w = torch.randn(o, i, k, k, device=DEV, generator=g) / math.sqrt(i * k * k)
    geo = conv_ops.Geometry("conv", k, k, 1, k // 2, (r, r), False)

    def cl(t):
        return t.contiguous(memory_format=torch.channels_last)

    def run(xs, gys, want_gw):
        xs = xs.detach().requires_grad_(True)
        wd = w.detach().requires_grad_(want_gw)
        y = conv_ops._ConvF.apply(xs, wd, None, geo)
        grads = torch.autograd.grad(y, (xs, wd) if want_gw else (xs,), gys)
        return y.detach(), grads[0].detach(), (grads[1].detach() if want_gw else None)

    gy = torch.randn(b, o, r, r, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)
    y, gx, gw = run(x, gy, with_wgrad)
    for lo, hi in ((0, 2), (b - 2, b)):
        ys, gxs, _ = run(cl(x[lo:hi]), cl(gy[lo:hi]), False)
        assert rel_err(y[lo:hi].float(), ys.float()) < 1e-2, f"forward, samples {lo}..{hi}"
        assert rel_err(gx[lo:hi].float(), gxs.float()) < 1e-2, f"data gradient, samples {lo}..{hi}"
    if with_wgrad:
        gw_parts = torch.zeros_like(gw)
        for lo, hi in ((0, 11), (11, 22), (22, b)):
            gw_parts += run(cl(x[lo:hi]), cl(gy[lo:hi]), True)[2]
        assert rel_err(gw, gw_parts) < 1e-3, "weight gradient"


@pytest.mark.parametrize("g,b,l,n,k", [(5, 16, 14, 512, 512), (3, 3, 4, 40, 24), (1, 1, 1, 8, 8)])
def test_grouped_linear_matches_per_layer(g, b, l, n, k):
    """conv_ops.grouped_linear (all style affines in one launch) == the per-layer EqualizedLinear path: values,
    first-order gradients (latent slots shared by several layers accumulate), and the second-order fallback."""
    from multi_stylegan_amd import conv_ops
    torch.manual_seed(3)
    latent = torch.randn(b, l, k, device=DEV, requires_grad=True)
    slots = [int(v) for v in torch.randint(0, l, (g,))]
    ws = [torch.randn(n, k, device=DEV, requires_grad=True) for _ in range(g)]
    bs = [torch.randn(n, device=DEV, requires_grad=True) for _ in range(g)]
    gy = torch.randn(g, b, n, device=DEV)
    wscale, bscale = 0.37, 1.3

    def per_layer():
        return torch.stack([conv_ops.linear(latent[:, slots[j]], ws[j], bs[j], wscale, bscale) for j in range(g)])

    out_g = conv_ops.grouped_linear(latent, slots, ws, bs, wscale, bscale)
    out_p = per_layer()
    assert rel_err(out_g, out_p) < 1e-6
    grads_g = torch.autograd.grad(out_g, [latent, *ws, *bs], gy)
    grads_p = torch.autograd.grad(out_p, [latent, *ws, *bs], gy)
    for a, c in zip(grads_g, grads_p):
        assert rel_err(a, c) < 1e-5
    # second order (path-length style): d/dW and d/d(cotangent) of |d out / d latent|^2 -- the latent-only first pass runs on
    # the grouped kernels' own differentiable node (_GroupedLinD, round 5) ...
    gyr = gy.clone().requires_grad_(True)
    def second(fn):
        o = fn()
        gl, = torch.autograd.grad(o, latent, gyr, create_graph=True)
        return torch.autograd.grad(gl.square().sum(), [*ws, gyr])
    for a, c in zip(second(lambda: conv_ops.grouped_linear(latent, slots, ws, bs, wscale, bscale)), second(per_layer)):
        assert rel_err(a, c) < 1e-5
    # ... and a first pass that also wants weight gradients takes the per-layer composite
    def second_full(fn):
        o = fn()
        gl, gw0 = torch.autograd.grad(o, [latent, ws[0]], gyr, create_graph=True)
        return torch.autograd.grad(gl.square().sum() + gw0.square().sum(), [ws[-1], gyr])
    for a, c in zip(second_full(lambda: conv_ops.grouped_linear(latent, slots, ws, bs, wscale, bscale)), second_full(per_layer)):
        assert rel_err(a, c) < 1e-5


@pytest.mark.parametrize("shape", [(4, 512, 512, 64, 64, 3, True), (2, 256, 256, 128, 128, 3, False),
                                    (2, 128, 512, 32, 256, 3, True), (2, 192, 512, 64, 256, 3, True), (4, 256, 256, 48, 48, 1, False),
                                    (2, 128, 128, 128, 128, 3, False), (8, 64, 256, 16, 512, 3, True),
                                    (8, 128, 128, 128, 256, 3, False), (16, 64, 384, 64, 64, 3, False),
                                    (32, 768, 768, 32, 32, 3, False)])
def test_conv_kernels_are_race_free(shape):
    """The forward kernels have no atomics: repeated launches on the same operands must agree bit for bit, and a
    staging race (an LDS-DMA piece still in flight when another wave reads it) shows up as a difference.  Shapes of the
    row-sharing kernel, the ping-pong kernel and the 128x128 LDS-DMA kernel, at sizes that fill the chip."""
    from multi_stylegan_amd import conv_ops
    b, i, o, h, w_, k, per_sample = shape
    torch.manual_seed(b * 1000 + i)
    x = conv_ops.to_compute_layout(torch.randn(b, i, h, w_, device=DEV), torch.bfloat16)
    w = torch.randn((b, o, i, k, k) if per_sample else (o, i, k, k), device=DEV) / math.sqrt(i * k * k)
    geo = conv_ops.Geometry("conv", k, k, 1, k // 2, (h, w_), per_sample)
    first = conv_ops._f_raw(x, w, None, geo).clone()
    for _ in range(12):
        again = conv_ops._f_raw(x, w, None, geo)
        assert torch.equal(first, again)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_equalized_transposed_conv_and_conv1d_golden(golden, dtype):
    """EqualizedTransposedConv2d (kernel 2, stride 2: the reference's defaults) and EqualizedConv1d (reference
    equalized_layer.py:77-207; not instantiated by G / D) against golden vectors from the reference's classes: output,
    input gradient, weight gradient.  Other transposed-conv geometries are refused, not approximated."""
    from multi_stylegan_amd import _lib, equalized_layer as E
    z = golden("layers")
    tol = TOLS[dtype] * (3 if dtype == torch.float32 else 1)
    cases = (("eqconvT2x2", E.EqualizedTransposedConv2d(6, 10, kernel_size=2, stride=2, padding=0)),
             ("eqconv1d", E.EqualizedConv1d(6, 10, kernel_size=3, stride=1, padding=1)),
             ("eqconv1d_s2", E.EqualizedConv1d(6, 10, kernel_size=5, stride=2, padding=2)))
    for name, mod in cases:
        assert float(mod.bias.detach()[0]) == 1.0                       # the reference initialises these biases to ones
        with torch.no_grad():
            mod.weight.copy_(z[name + ".w"]); mod.bias.copy_(z[name + ".b"])
        mod.to(DEV)
        x = z[name + ".x"].to(DEV, dtype).requires_grad_(True)
        y = mod(x)
        assert y.shape == z[name + ".y"].shape
        gx, gw = torch.autograd.grad(y, (x, mod.weight), z[name + ".gy"].to(DEV, dtype))
        assert rel_err(y.float(), z[name + ".y"]) < tol, name
        assert rel_err(gx.float(), z[name + ".gx"]) < tol, name
        assert rel_err(gw.float(), z[name + ".gw"]) < tol, name
    with pytest.raises(_lib.MsgHipError, match="only kernel_size 2"):
        E.EqualizedTransposedConv2d(6, 10, kernel_size=3, stride=2, padding=1).to(DEV)(torch.zeros(1, 6, 4, 4, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["conv3x3_demod", "up2x2_demod", "torgb1x1_nodemod", "conv3x3_nodemod"])
def test_modulated_conv_second_order_native_matches_composite(kind, dtype):
    """The path-length pattern -- a first backward under create_graph, then a scalar of its style AND input gradients
    differentiated with respect to everything -- on the native second-order node (conv_ops._ModConvGrad: msg_scale_rows_cols2,
    msg_modulate_backward2 and the contraction kernels) against the composite torch-op graph it replaces, and against an
    fp64 evaluation of the reference's formulation (multi_stylegan_generator.py:384-411)."""
    from multi_stylegan_amd import conv_ops
    torch.manual_seed(hash(kind) % 1000)
    b, i, o, h = 5, 72, (3 if "torgb" in kind else 40), 12
    k = 3 if "3x3" in kind else (2 if "up2" in kind else 1)
    demod, up = "nodemod" not in kind, "up2" in kind
    x0 = torch.randn(b, i, h, h, device=DEV)
    w0 = torch.randn(1, o, i, k, k, device=DEV)
    s0 = torch.randn(b, i, device=DEV) * 0.5 + 1.0
    oh = 2 * h if up else h
    gy0 = torch.randn(b, o, oh, oh, device=DEV)
    r_x = torch.randn(b, i, h, h, device=DEV)

    def run(native, dt):
        conv_ops._NATIVE_SECOND_ORDER = native
        try:
            x = conv_ops.to_compute_layout(x0.clone(), dt).requires_grad_(True)
            w = torch.nn.Parameter(w0.clone())
            s = s0.clone().requires_grad_(True)
            gy = conv_ops.to_compute_layout(gy0.clone(), dt).requires_grad_(True)
            y = conv_ops.modulated_conv2d(x, w, s, demod, up)
            gx, gs = torch.autograd.grad(y, (x, s), gy, create_graph=True)
            pen = gs.float().square().sum() + (gx.float() * r_x).sum()
            return [t.float() for t in torch.autograd.grad(pen, (x, w, s, gy))] + [gs.detach().float(), gx.detach().float()]
        finally:
            conv_ops._NATIVE_SECOND_ORDER = True

    def reference64():
        x, w, s, gy = (t.double().cpu().requires_grad_(True) for t in (x0, w0, s0, gy0))
        scale = math.sqrt(2.0) / math.sqrt(i * k * k)
        wm = scale * w * s[:, None, :, None, None]
        if demod:
            wm = wm * torch.rsqrt(wm.pow(2).sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
        if up:
            y = torch.nn.functional.conv_transpose2d(x.reshape(1, b * i, h, h), wm.transpose(1, 2).reshape(b * i, o, k, k),
                                                     stride=2, groups=b).reshape(b, o, oh, oh)
        else:
            y = torch.nn.functional.conv2d(x.reshape(1, b * i, h, h), wm.reshape(b * o, i, k, k), padding=k // 2,
                                           groups=b).reshape(b, o, oh, oh)
        gx, gs = torch.autograd.grad(y, (x, s), gy, create_graph=True)
        pen = gs.square().sum() + (gx * r_x.double().cpu()).sum()
        return list(torch.autograd.grad(pen, (x, w, s, gy))) + [gs.detach(), gx.detach()]

    calls = []
    orig = conv_ops._ModConvGrad.backward
    conv_ops._ModConvGrad.backward = staticmethod(lambda ctx, *g: (calls.append(1), orig(ctx, *g))[1])
    try:
        native = run(True, dtype)
    finally:
        conv_ops._ModConvGrad.backward = orig
    assert calls, "the native second-order node did not run"
    composite = run(False, dtype)
    names = ("d/dx", "d/dW", "d/ds", "d/dgy", "gs", "gx")
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    for name, a, c in zip(names, native, composite):
        assert rel_err(a, c) < tol, (name, rel_err(a, c))
    if dtype == torch.float32:
        for name, a, r in zip(names, native, reference64()):
            assert rel_err(a, r) < 2e-4, (name, rel_err(a, r))


@pytest.mark.parametrize("batch,hw", [(2, (128, 128)), (8, (128, 128)), (3, (136, 124)), (1, (256, 256))])
def test_activation_stationary_upconv_kernel(batch, hw):
    """conv_upconv.hip -- the generator's sub-pixel up-convolution (512 -> 4 x 512, per-sample weights, pixel-shuffled output) on
    its own activation-stationary kernel: against the transposed convolution it implements (fp32 on the same bf16-rounded
    operands), bit-identical over repeated launches, including a map whose pixel count is not a multiple of the 128-pixel
    tile and both workgroup orders (samples dealt to XCDs when the batch is a multiple of 8, sample-major otherwise)."""
    from multi_stylegan_amd import _lib, conv_ops
    h, w_ = hw
    i = o = 512
    torch.manual_seed(batch + h)
    x = conv_ops.to_compute_layout(torch.randn(batch, i, h, w_, device=DEV), torch.bfloat16)
    w = (torch.randn(batch, o, i, 2, 2, device=DEV) / math.sqrt(i)).bfloat16().float()
    geo = conv_ops.Geometry("up2", 2, 2, 1, 0, (h, w_), True)
    wk, ck = conv_ops._relay_fwd_kind(w, torch.bfloat16, "up2")
    assert _lib.lib().msg_conv2d_fprop_upconv_eligible(batch, h, w_, i, ck, h, w_, 4 * o, 1, 1, 1, 0, 1, 1, wk.stride(0)) == 1
    y = conv_ops._f_raw(x, w, None, geo)
    assert y.shape == (batch, o, 2 * h, 2 * w_)
    for _ in range(4):
        assert torch.equal(y, conv_ops._f_raw(x, w, None, geo))
    want = torch.cat([torch.nn.functional.conv_transpose2d(x[s:s + 1].float(), w[s].transpose(0, 1), stride=2)
                      for s in range(batch)])
    assert rel_err(y.float(), want) < 1e-2                      # (bf16 output rounding; the accumulation is fp32)
    # gradients still flow through the existing data / weight gradient kernels of the geometry
    xg = x.detach().clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    yy = conv_ops._ConvF.apply(xg, wg, None, geo)
    gy = conv_ops.to_compute_layout(torch.randn_like(want), torch.bfloat16)
    gx, gw = torch.autograd.grad(yy, (xg, wg), gy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = torch.cat([torch.nn.functional.conv_transpose2d(xr[s:s + 1], wr[s].transpose(0, 1), stride=2) for s in range(batch)])
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.float())
    assert rel_err(gx.float(), gxr) < 2e-2 and rel_err(gw.float(), gwr) < 2e-2


@pytest.mark.parametrize("case", [
    # (batch, in, out, (h, w), per-sample weights, bias, residual merge)
    (2, 512, 6, (64, 64), True, True, False),       # the generator's RGB heads (two streams, 3 + 3)
    (2, 128, 1, (96, 80), False, False, False),     # the discriminator's pixel-wise head
    (3, 128, 6, (67, 61), False, True, False),      # ragged pixel count
    (2, 256, 8, (64, 64), True, False, False),
    (2, 6, 512, (64, 64), True, False, False),      # data gradient of the RGB heads
    (2, 6, 128, (96, 80), False, False, True),      # the first block's residual conv with the merge in its epilogue
    (3, 1, 128, (67, 61), False, True, False),      # data gradient of the pixel-wise head, ragged
    (2, 3, 64, (64, 64), False, False, False),
])
def test_thin_pointwise_conv_kernels(case):
    """conv_thin.hip -- 1x1 convolutions with <= 8 channels on one side as streaming kernels: against fp32 torch on the same
    bf16-rounded operands, bit-identical over repeated launches, pad channels of the output left zero."""
    from multi_stylegan_amd import _lib, conv_ops
    batch, i, o, (h, w_), per_sample, with_bias, with_res = case
    torch.manual_seed(i * 7 + o)
    x = conv_ops.to_compute_layout(torch.randn(batch, i, h, w_, device=DEV), torch.bfloat16)
    shape = (batch, o, i, 1, 1) if per_sample else (o, i, 1, 1)
    w = (torch.randn(*shape, device=DEV) / math.sqrt(i)).bfloat16().float()
    bias = torch.randn(o, device=DEV) if with_bias else None
    geo = conv_ops.Geometry("conv", 1, 1, 1, 0, (h, w_), per_sample)
    xv, cx = conv_ops._nhwc_view(x)
    ck = 64 * ((i + 63) // 64)
    mode = _lib.lib().msg_conv2d_fprop_thin_eligible(batch, h, w_, cx, ck, h, w_, o, 8 * ((o + 7) // 8), 1, 1, 1, 0, 1, 0,
                                                     2 if with_res else 0)
    assert mode == (1 if o <= 8 else 2)                                   # (the kernels under test do run)
    res = conv_ops.to_compute_layout(torch.randn(batch, o, h, w_, device=DEV), torch.bfloat16) if with_res else None
    y = conv_ops._f_raw(x, w, bias, geo, residual=(res, 0.5) if with_res else None)
    for _ in range(3):
        assert torch.equal(y, conv_ops._f_raw(x, w, bias, geo, residual=(res, 0.5) if with_res else None))
    if per_sample:
        want = torch.cat([torch.nn.functional.conv2d(x[s:s + 1].float(), w[s], bias) for s in range(batch)])
    else:
        want = torch.nn.functional.conv2d(x.float(), w, bias)
    if with_res:
        want = (want.bfloat16().float() + res.float()) * 0.5
    assert y.shape == want.shape and rel_err(y.float(), want) < 6e-3      # (bf16 output rounding; fp32 accumulation)
    if o % 8:                                                             # the padded channel-vector behind the real outputs
        base = y._base if y._base is not None else y
        assert base.shape[-1] == 8 * ((o + 7) // 8) or base.shape[1] == 8 * ((o + 7) // 8)


@pytest.mark.parametrize("shape", [(3, 6, 40, 64), (2, 6, 5, 32), (1, 3, 33, 96), (2, 7, 16, 128), (2, 6, 12, 20)])
def test_gather_taps_kernels(shape):
    """msg_gather_taps (the 3x3 taps of a <= 8-channel input laid out as ONE 128-byte K run per pixel, the form the
    discriminator's first layer contracts): the LDS-staged kernel for map widths that are multiples of 32 and the generic
    one (last shape) against the definition, exactly; and the conv built on it against torch."""
    from multi_stylegan_amd import conv_ops
    b, c, h, w_ = shape
    torch.manual_seed(h)
    x = torch.randn(b, c, h, w_, device=DEV).bfloat16()
    geo = conv_ops.Geometry("conv", 3, 3, 1, 1, (h, w_), False)
    got, ko = conv_ops._gather_taps(conv_ops.to_compute_layout(x), c, geo)
    assert ko == 64 and got.shape == (b, 64, h, w_)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    want = torch.zeros(b, 64, h, w_, device=DEV, dtype=torch.bfloat16)
    for t in range(9):
        want[:, t * c:(t + 1) * c] = xp[:, :, t // 3:t // 3 + h, t % 3:t % 3 + w_]
    assert torch.equal(got, want)
    wgt = (torch.randn(16, c, 3, 3, device=DEV) / 7).bfloat16().float()
    y = conv_ops.conv2d(conv_ops.to_compute_layout(x), wgt, padding=1)
    assert rel_err(y.float(), torch.nn.functional.conv2d(x.float(), wgt, padding=1)) < 1e-2


@pytest.mark.parametrize("shape", [(3, 6, 40, 64, 128), (2, 6, 5, 32, 64), (1, 3, 33, 96, 72), (2, 7, 16, 20, 128)])
def test_few_channel_data_gradient_through_the_fold(shape):
    """Data gradient of a 'same' 3x3 conv over <= 8 channels (the discriminator's first layer, 128 -> 6 channels @256^2) as a
    1x1 contraction to the 9 x C tap planes + msg_fold_taps (conv_ops._d_raw_thin): against the fp64 definition, with and
    without the map the residual epilogue adds, through autograd (the forward is the tap-gathered form), and the fold alone
    against its definition."""
    from multi_stylegan_amd import conv_ops
    b, c, h, w_, o = shape
    torch.manual_seed(h + o)
    wgt = (torch.randn(o, c, 3, 3, device=DEV) / math.sqrt(9 * c)).requires_grad_(True)
    gy = torch.randn(b, o, h, w_, device=DEV).bfloat16()
    other = torch.randn(b, c, h, w_, device=DEV).bfloat16()
    geo = conv_ops.Geometry("conv", 3, 3, 1, 1, (h, w_), False, 0.7)
    assert conv_ops._thin_ok(torch.bfloat16, c, geo)
    want = F.conv_transpose2d(gy.double(), 0.7 * wgt.detach().double().bfloat16().double(), padding=1)
    got = conv_ops._d_raw(conv_ops.to_compute_layout(gy), wgt, geo)
    assert got.shape == (b, c, h, w_) and rel_err(got.float(), want) < 8e-3
    got2 = conv_ops._d_raw(conv_ops.to_compute_layout(gy), wgt, geo, residual=(conv_ops.to_compute_layout(other), 1.0))
    assert rel_err(got2.float(), want + other.double()) < 8e-3
    # the padding channels of the 16-byte pixel vector are zero (the next reader takes whole vectors)
    base, cx = conv_ops._nhwc_view(got)
    assert cx == 8 and float(base.permute(0, 2, 3, 1).as_strided((b, h, w_, 8), (h * w_ * 8, w_ * 8, 8, 1))[..., c:].abs().max()) == 0.0
    # through autograd: forward (tap gather + 1x1) and backward (1x1 + fold) against torch's conv
    x = torch.randn(b, c, h, w_, device=DEV).bfloat16()
    xd = conv_ops.to_compute_layout(x).detach().requires_grad_(True)
    y = conv_ops.conv2d(xd, wgt, padding=1, wscale=0.7)
    gx, = torch.autograd.grad(y, xd, conv_ops.to_compute_layout(gy))
    assert rel_err(gx.float(), want) < 8e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(32, 128, 127, 127), (3, 24, 5, 7), (2, 768, 15, 15)])
def test_channel_sums(shape, dtype):
    """msg_channel_sums -- the bias gradient of a conv with no activation behind it -- against the library reduction in fp64,
    bit-identical over repeated launches; layouts it does not take go to the library."""
    from multi_stylegan_amd import conv_ops
    torch.manual_seed(shape[1])
    if shape[0] * shape[1] * shape[2] * shape[3] > 4e7:
        shape = (8,) + shape[1:]
    g = torch.randn(*shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    a, b = conv_ops._channel_sums(g), conv_ops._channel_sums(g)
    assert a.dtype == torch.float32 and torch.equal(a, b)
    want = g.double().sum(dim=(0, 2, 3))
    assert rel_err(a.double(), want) < 1e-5
    assert rel_err(conv_ops._channel_sums(g.contiguous()).double(), want) < 1e-5          # (NCHW: library path)



@pytest.mark.parametrize("b,c,hw,n_head,other,noise_batch", [(2, 512, 32, 6, True, 2), (3, 128, 16, 6, False, 1),
                                                             (2, 384, 8, 3, True, 0), (1, 512, 64, 8, False, 1)])
def test_activation_backward_forms_the_head_gradient(b, c, hw, n_head, other, noise_batch):
    """msg_bias_act_backward_mask_head against its definition in fp32 torch ops: gx = (gy + h) * scale * slope with
    h[q][c] = wscale * style[b][c] * sum_o ghead[q][o] * W[o][c], slope from the sign bytes; bias / noise-weight sums of the
    UNROUNDED gx.  bf16 head weights and one bf16 rounding of gx (2^-7 of max|gx| per element), sums at 2e-3; garbage in the padding planes of the head's
    gradient must not reach the result."""
    from multi_stylegan_amd.op_static import fused_act
    torch.manual_seed(b * 1000 + c + hw)
    bf = torch.bfloat16
    y = torch.randn(b, c, hw, hw, device=DEV)
    pos = (y > 0)
    bits = pos.permute(0, 2, 3, 1).reshape(b * hw * hw, c // 8, 8).to(torch.int32)
    mbytes = (bits * (1 << torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous()
    gy = torch.randn(b, c, hw, hw, device=DEV).to(bf).contiguous(memory_format=torch.channels_last) if other else None
    hbuf = torch.full((b, hw, hw, 8), float("nan"), device=DEV, dtype=bf)             # padding planes: NaN
    hbuf[..., :n_head] = torch.randn(b, hw, hw, n_head, device=DEV).to(bf)
    ghead = hbuf.permute(0, 3, 1, 2)[:, :n_head]
    whead = torch.randn(n_head, c, device=DEV)
    style = 1 + 0.3 * torch.randn(b, c, device=DEV)
    wscale, alpha, scale = 0.044, 0.2, math.sqrt(2)
    noise = torch.randn(noise_batch, 1, hw, hw, device=DEV) if noise_batch else None
    bias = torch.zeros(c, device=DEV)
    got = fused_act.act_backward_with_head(gy, (ghead, whead, style, wscale), (b, c, hw, hw), noise, None, True, alpha, scale,
                                           (mbytes, 1, c))
    assert got is not None
    gx, gb, gnw = got
    h = torch.einsum("bohw,oc,bc->bchw", ghead.float(), whead, style) * wscale
    ref = ((gy.float() if other else 0) + h) * scale * torch.where(pos, 1.0, alpha)
    assert gx.dtype == bf and gx.is_contiguous(memory_format=torch.channels_last)
    err = (gx.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2 ** -7, err            # (the per-sample head weights enter the MFMA rounded to bf16, then one rounding of gx)
    assert rel_err(gb, ref.sum(dim=(0, 2, 3))) < 2e-3
    if noise is not None:
        assert rel_err(gnw, (ref * noise).sum().reshape(1)) < 5e-3
    else:
        assert gnw.numel() == 0


@pytest.mark.parametrize("batch", [8, 4])
def test_image_head_gradient_handed_to_the_producer(batch, monkeypatch):
    """(batch 4: the producer runs on a kernel that leaves no sign bytes -- the hand-over happens, the fused pass declines and
    the producer takes the head's data gradient the ordinary way: same tolerances, no fused launch.)
    A styled 3x3 layer (fused activation, sign bytes) whose output feeds a second consumer and a 6-plane image head: with
    the hand-over (conv_ops.HeadGradSlot) the head's data gradient is formed inside the layer's activation backward; every
    leaf gradient agrees with the form that writes it as a map and lets autograd add (bf16 roundings of two intermediate maps
    apart: 1e-2 norm-wise on the bf16 input gradient, 5e-3 on the fp32 sums), the fused kernel is the one that ran; without a second consumer the producer sees no incoming gradient at all."""
    from multi_stylegan_amd import _lib, conv_ops
    bf = torch.bfloat16

    def run(fuse, second_consumer):
        monkeypatch.setattr(conv_ops, "HEAD_GRAD_FUSION", fuse)
        torch.manual_seed(5)
        x = conv_ops.to_compute_layout(torch.randn(batch, 512, 64, 64, device=DEV), bf).requires_grad_(True)   # (8: the row-sharing kernel)
        w = torch.randn(1, 512, 512, 3, 3, device=DEV).requires_grad_(True)
        style = (1 + 0.1 * torch.randn(batch, 512, device=DEV)).requires_grad_(True)
        bias = (0.1 * torch.randn(512, device=DEV)).requires_grad_(True)
        noise = torch.randn(batch, 1, 64, 64, device=DEV)
        nw = torch.full((1,), 0.3, device=DEV, requires_grad=True)
        wh = torch.randn(1, 6, 512, 1, 1, device=DEV).requires_grad_(True)
        sh = (1 + 0.1 * torch.randn(batch, 512, device=DEV)).requires_grad_(True)
        slot = conv_ops.HeadGradSlot()
        y = conv_ops.modulated_conv2d_bias_act(x, w, style, True, bias, noise, nw, scale=math.sqrt(2), head_slot=slot)
        rgb = conv_ops.modulated_conv2d(y, wh, sh, False, False, head_slot=slot)
        g_rgb = torch.randn(rgb.shape, device=DEV)
        loss = (rgb.float() * g_rgb).sum()
        if second_consumer:
            g_y = torch.randn(y.shape, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
            loss = loss + (y * g_y).float().sum()
        _lib.kernel_clock.reset(enabled=True)
        grads = torch.autograd.grad(loss, (x, w, style, bias, nw, wh, sh))
        torch.cuda.synchronize()
        keys = set(_lib.kernel_clock.summary())
        _lib.kernel_clock.reset(enabled=False)
        return grads, keys

    for second in (True, False):
        g1, k1 = run(True, second)
        g0, k0 = run(False, second)
        assert any(k.startswith("bias_act_bwd_mask_head/") for k in k1) == (batch == 8), k1
        assert not any(k.startswith("bias_act_bwd_mask_head/") for k in k0)
        for name, a, r in zip(("x", "w", "style", "bias", "noise_w", "w_head", "style_head"), g1, g0):
            e = ((a.float() - r.float()).norm() / r.float().norm()).item()
            assert e < (1e-2 if a.dtype == bf else 2e-2 if name == "noise_w" else 5e-3), (name, second, e)   # (noise_w: one cancelling sum)


@pytest.mark.parametrize("case", ["128ch_small_tile", "256ch_large_tile", "sign_from_the_map", "with_residual"])
def test_activation_backward_in_the_data_gradient_epilogue(case, monkeypatch):
    """conv -> bias + leaky ReLU -> conv (the main path of a discriminator block): with the hand-over (conv_ops.ActHandle) the
    second conv's data-gradient launch applies the first activation's backward in its epilogue (msg_conv2d_fprop_act_backward)
    and leaves the bias gradient; the map between the two backward nodes is not written.  Same arithmetic on the same rounded
    values as the two-pass form: input and weight gradients bit for bit, the bias gradient (another summation order) to 1e-5.
    Sign source: the first conv's sign bytes (tiles of 128 / 256 pixels) or, when no bytes were written, its stored output;
    with_residual: the second conv's input has a second gradient that meets the data gradient in the same epilogue."""
    from multi_stylegan_amd import _lib, conv_ops
    from multi_stylegan_amd.op_static import fused_act
    bf = torch.bfloat16
    b, c, hw = (16, 256, 128) if case == "256ch_large_tile" else (4, 128, 128)

    def run(fuse):
        monkeypatch.setattr(conv_ops, "ACT_BACKWARD_IN_DGRAD", fuse)
        monkeypatch.setattr(fused_act, "ACT_MASK", case != "sign_from_the_map")
        torch.manual_seed(17)
        x = conv_ops.to_compute_layout(torch.randn(b, c, hw, hw, device=DEV), bf).requires_grad_(True)
        w1 = (torch.randn(c, c, 3, 3, device=DEV) / math.sqrt(9 * c)).requires_grad_(True)
        w2 = (torch.randn(c, c, 3, 3, device=DEV) / math.sqrt(9 * c)).requires_grad_(True)
        b1 = (0.1 * torch.randn(c, device=DEV)).requires_grad_(True)
        b2 = (0.1 * torch.randn(c, device=DEV)).requires_grad_(True)
        h = conv_ops.ActHandle()
        slot = conv_ops.GradSlot() if case == "with_residual" else None
        y1 = conv_ops.conv2d_bias_act(x, w1, b1, padding=1, scale=math.sqrt(2), act_handle=h)
        if slot is not None:
            y1, y1_side = conv_ops.fork_input(y1, slot)
        y2 = conv_ops.conv2d_bias_act(y1, w2, b2, padding=1, scale=math.sqrt(2), input_act=h, grad_slot=slot)
        gy = torch.randn(y2.shape, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
        loss = (y2 * gy).float().sum()
        if slot is not None:
            loss = loss + _SideConsumer.apply(y1_side, slot).float().sum()
        _lib.kernel_clock.reset(enabled=True)
        grads = torch.autograd.grad(loss, (x, w1, w2, b1, b2))
        torch.cuda.synchronize()
        keys = set(_lib.kernel_clock.summary())
        _lib.kernel_clock.reset(enabled=False)
        return grads, keys

    g1, k1 = run(True)
    g0, k0 = run(False)
    assert any("_actbwd" in k for k in k1), k1
    assert not any("_actbwd" in k for k in k0)
    for name, a, r in zip(("x", "w1", "w2", "b1", "b2"), g1, g0):
        if name == "b1":
            assert rel_err(a, r) < 1e-5, name
        else:
            assert torch.equal(a, r), (name, rel_err(a, r))


class _SideConsumer(torch.autograd.Function):
    """A second consumer of a forked map that leaves its gradient in the fork's slot (as a block's 1x1 residual conv does)."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        ctx.slot.g = g.contiguous(memory_format=torch.channels_last)
        return ctx.slot.g, None


def test_blur_activation_backward_in_the_styled_conv_data_gradient(monkeypatch):
    """blur -> noise + bias + leaky ReLU (the tail of the generator's upsampling layer) -> styled 3x3 conv with per-sample weights:
    the conv's data-gradient launch (the 256 x 256 tile, K = 9 x 512) applies the blur stage's activation backward in its
    epilogue from the stage's plain-layout sign bytes and leaves its bias and noise-weight gradients (conv_ops.ActHandle).
    Input / weight / style gradients bit for bit the two-pass form's, the two sums to 1e-5."""
    from multi_stylegan_amd import _lib, conv_ops
    from multi_stylegan_amd.op_static import blur_bias_act
    bf = torch.bfloat16

    def run(fuse):
        monkeypatch.setattr(conv_ops, "ACT_BACKWARD_IN_DGRAD", fuse)
        torch.manual_seed(23)
        x = conv_ops.to_compute_layout(torch.randn(8, 512, 67, 67, device=DEV), bf).requires_grad_(True)
        fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 16).to(DEV)
        b1 = (0.1 * torch.randn(512, device=DEV)).requires_grad_(True)
        n1 = torch.randn(8, 1, 64, 64, device=DEV)
        nw1 = torch.full((1,), 0.3, device=DEV, requires_grad=True)
        w = torch.randn(1, 512, 512, 3, 3, device=DEV).requires_grad_(True)
        style = (1 + 0.1 * torch.randn(8, 512, device=DEV)).requires_grad_(True)
        b2 = (0.1 * torch.randn(512, device=DEV)).requires_grad_(True)
        n2 = torch.randn(8, 1, 64, 64, device=DEV)
        nw2 = torch.full((1,), 0.2, device=DEV, requires_grad=True)
        h = conv_ops.ActHandle()
        y1 = blur_bias_act(x, fir, (0, 0), b1, n1, nw1, scale=math.sqrt(2), act_handle=h)
        y2 = conv_ops.modulated_conv2d_bias_act(y1, w, style, True, b2, n2, nw2, scale=math.sqrt(2), input_act=h)
        gy = torch.randn(y2.shape, device=DEV).to(bf).contiguous(memory_format=torch.channels_last)
        _lib.kernel_clock.reset(enabled=True)
        grads = torch.autograd.grad((y2 * gy).float().sum(), (x, w, style, b2, nw2, b1, nw1))
        torch.cuda.synchronize()
        keys = set(_lib.kernel_clock.summary())
        _lib.kernel_clock.reset(enabled=False)
        return grads, keys

    g1, k1 = run(True)
    g0, k0 = run(False)
    assert any("_actbwd" in k for k in k1), k1
    assert not any("_actbwd" in k for k in k0)
    for name, a, r in zip(("x", "w", "style", "b2", "nw2", "b1", "nw1"), g1, g0):
        if name in ("b1", "nw1"):
            assert rel_err(a, r) < 1e-5, (name, rel_err(a, r))
        else:
            assert torch.equal(a, r), (name, rel_err(a, r))


@pytest.mark.parametrize("b,c,hw", [(4, 128, 64), (2, 64, 33), (3, 512, 16)])
def test_pixel_wise_head_one_pass(b, c, hw, monkeypatch):
    """FusedLeakyReLU(C) -> 1x1 EqualizedConv2d(C, 1) as one streaming pass per direction (msg_act_pointwise_head) against
    the two-op form in fp32 torch ops on the same bf16 map: output 1e-3 (fp32 inside the pass), input gradient one bf16
    rounding, bias / weight gradients 1e-3; a differentiated backward (the R1 regulariser's) goes through the two-op form."""
    from multi_stylegan_amd import equalized_layer
    from multi_stylegan_amd.op_static import FusedLeakyReLU, pointwise_head
    torch.manual_seed(c + hw)
    act = FusedLeakyReLU(c).to(DEV)
    conv = equalized_layer.EqualizedConv2d(c, 1, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=False).to(DEV)
    with torch.no_grad():
        act.bias.copy_(0.3 * torch.randn(c))
    x = torch.randn(b, c, hw, hw, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert pointwise_head.supported(x, conv, act)
    y = pointwise_head.act_pointwise_head(x, act, conv)
    gy = torch.randn(y.shape, device=DEV)
    gx, gb, gw = torch.autograd.grad(y, (x, act.bias, conv.weight), gy)
    xf = x.detach().float().requires_grad_(True)
    bf_, wf = act.bias.detach().clone().requires_grad_(True), conv.weight.detach().clone().requires_grad_(True)
    a = torch.nn.functional.leaky_relu(xf + bf_.view(1, -1, 1, 1), act.negative_slope) * act.scale
    ref = torch.nn.functional.conv2d(a, wf * conv.scale)
    rgx, rgb, rgw = torch.autograd.grad(ref, (xf, bf_, wf), gy)
    assert y.dtype == torch.float32 and rel_err(y, ref) < 1e-3
    assert rel_err(gx, rgx) < 2 ** -8 and rel_err(gb, rgb) < 1e-3 and rel_err(gw, rgw) < 1e-3
    # second order: d/dx of sum(gx_created ** 2) exists and matches the two-op form's
    x2 = x.detach().clone().requires_grad_(True)
    y2 = pointwise_head.act_pointwise_head(x2, act, conv)
    g1, = torch.autograd.grad(y2.sum(), x2, create_graph=True)
    (g1.float().square().sum()).backward()
    assert act.bias.grad is not None or conv.weight.grad is not None
    monkeypatch.setattr(pointwise_head, "POINTWISE_HEAD", False)
    assert not pointwise_head.supported(x, conv, act)
