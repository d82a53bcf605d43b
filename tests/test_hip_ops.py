"""Parity of the gfx950 kernels (called through the C ABI via multi_stylegan_amd.op_static) against the golden
vectors and the CPU oracle.  fp32 tolerance: 1e-3 relative (BASELINE.json north_star); in practice ~1e-6.
bf16 storage: 2e-2 relative to max|ref| (8-bit mantissa on inputs and outputs, fp32 arithmetic inside)."""
import math

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL32 = 1e-3
TOL16 = 2e-2
DEV = "cuda:0"

UPFIRDN_CASES = ["g_blur_pad21_gain4", "g_skip_up2_pad21", "d_blur_pad22_odd", "bwd_of_up2_down2",
                 "asym_blur_pad21", "asym_up2_pad21", "asym_down2_pad12", "blur_pad11"]


def _ops():
    from multi_stylegan_amd import op_static
    return op_static


@pytest.mark.parametrize("case", UPFIRDN_CASES)
def test_upfirdn2d_golden_nchw(golden, case):
    z, ops = golden("upfirdn2d"), _ops()
    up, down, p0, p1 = [int(v) for v in z[case + ".cfg"]]
    x = z[case + ".x"].to(DEV).requires_grad_(True)
    gy = z[case + ".gy"].to(DEV).requires_grad_(True)
    fir = z[case + ".fir"].to(DEV)
    y = ops.upfirdn2d(x, fir, up=up, down=down, pad=(p0, p1))
    gx, = torch.autograd.grad(y, x, gy, create_graph=True)
    ggy, = torch.autograd.grad(gx, gy, z[case + ".ggx"].to(DEV))
    assert rel_err(y, z[case + ".y"]) < TOL32
    assert rel_err(gx, z[case + ".gx"]) < TOL32
    assert rel_err(ggy, z[case + ".ggy"]) < TOL32


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)])
@pytest.mark.parametrize("cfg", [  # (up, down, pad, channels, h, w)
    (1, 1, (2, 1), 16, 12, 10), (1, 1, (2, 2), 8, 15, 15), (2, 1, (2, 1), 24, 7, 9), (1, 2, (1, 1), 16, 14, 18),
    (2, 1, (1, 2), 8, 5, 6), (1, 2, (2, 1), 40, 9, 11), (1, 1, (1, 1), 12, 5, 5), (3, 2, (2, 3), 8, 6, 7),
    (1, 1, (-1, 0), 8, 9, 9)])
def test_upfirdn2d_channels_last_vs_oracle(cfg, dtype, tol):
    """The vectorised channels-last fast paths (and the generic one for odd configs) against the CPU oracle,
    forward + adjoint + second order, asymmetric random FIR so any flip/transposition shows."""
    from oracle import ops as oo
    ops = _ops()
    up, down, pad, c, h, w = cfg
    g = torch.Generator().manual_seed(hash(cfg) % 1000)
    fir = torch.randn(4, 4, generator=g) if up != 3 else torch.randn(5, 3, generator=g)
    x = torch.randn(2, c, h, w, generator=g).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    yr = oo.upfirdn2d(xr, fir, up=up, down=down, pad=pad)
    gy = torch.randn(yr.shape, generator=g).to(dtype).float()
    gyr = gy.clone().requires_grad_(True)
    gxr, = torch.autograd.grad(yr, xr, gyr, create_graph=True)
    ggx = torch.randn(x.shape, generator=g).to(dtype).float()
    ggyr, = torch.autograd.grad(gxr, gyr, ggx)
    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gyd = gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.upfirdn2d(xd, fir.to(DEV), up=up, down=down, pad=pad)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == dtype
    gx, = torch.autograd.grad(y, xd, gyd, create_graph=True)
    ggy, = torch.autograd.grad(gx, gyd, ggx.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    assert rel_err(y.float(), yr) < tol
    assert rel_err(gx.float(), gxr) < tol
    assert rel_err(ggy.float(), ggyr) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)])
@pytest.mark.parametrize("cfg", [((2, 1), 16, 12, 10, 2), ((2, 2), 8, 15, 15, 3), ((1, 1), 24, 5, 33, 1),
                                 ((2, 1), 64, 40, 19, 2), ((3, 3), 8, 3, 2, 1), ((0, 0), 8, 9, 9, 1),
                                 ((-1, 0), 16, 20, 20, 1)])
def test_blur_separable_kernel(cfg, dtype, tol):
    """up = down = 1 with a rank-1 4x4 FIR takes csrc/blur_sep.hip (sliding-window separable form).  Asymmetric
    factors so a flipped / transposed tap order shows; checked against the CPU oracle (forward, adjoint, second
    order) and against the 2-D kernel on the same data; ragged widths, rows beyond one 16-row strip, crops."""
    from oracle import ops as oo
    import importlib
    mod = importlib.import_module("multi_stylegan_amd.op_static.upfirdn2d")
    ops = _ops()
    pad, c, h, w, b = cfg
    g = torch.Generator().manual_seed(c * 100 + h)
    fy, fx = torch.randn(4, generator=g), torch.randn(4, generator=g)
    fir = torch.outer(fy, fx)
    x = torch.randn(b, c, h, w, generator=g).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    yr = oo.upfirdn2d(xr, fir, pad=pad)
    gy = torch.randn(yr.shape, generator=g).to(dtype).float()
    gyr = gy.clone().requires_grad_(True)
    gxr, = torch.autograd.grad(yr, xr, gyr, create_graph=True)
    ggx = torch.randn(x.shape, generator=g).to(dtype).float()
    ggyr, = torch.autograd.grad(gxr, gyr, ggx)
    fird = fir.to(DEV)
    assert mod._separable(fird) is not None and mod._separable(torch.randn(4, 4, generator=g).to(DEV)) is None
    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gyd = gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    saved = mod._SEPARABLE
    try:
        mod._SEPARABLE = 2                              # both storage types through the separable kernel
        y = ops.upfirdn2d(xd, fird, pad=pad)
        gx, = torch.autograd.grad(y, xd, gyd, create_graph=True)
        ggy, = torch.autograd.grad(gx, gyd, ggx.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
        mod._SEPARABLE = 0
        y2d = ops.upfirdn2d(xd, fird, pad=pad)
    finally:
        mod._SEPARABLE = saved
    assert rel_err(y.float(), yr) < tol
    assert rel_err(gx.float(), gxr) < tol
    assert rel_err(ggy.float(), ggyr) < tol
    assert rel_err(y.float(), y2d.float()) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("noise_batch", [1, 3])
@pytest.mark.parametrize("cfg", [((2, 1), 16, 12, 10), ((2, 2), 8, 15, 15), ((1, 1), 64, 40, 19)])
def test_blur_bias_act_matches_two_passes(cfg, noise_batch):
    """msg_upfirdn2d_separable_act vs msg_upfirdn2d_separable followed by msg_fused_bias_act (bf16 storage): equal up to
    the intermediate rounding of the blur result that only the two-pass form applies (<= ~1 bf16 ulp), and closer to the
    fp32 oracle; first- and second-order gradients through the same composed backward."""
    ops = _ops()
    from multi_stylegan_amd.op_static import blur_bias_act, fused_bias_noise_leaky_relu
    pad, c, h, w = cfg
    b = 3
    g = torch.Generator().manual_seed(c + h)
    fir = torch.outer(torch.randn(4, generator=g), torch.randn(4, generator=g)).to(DEV)
    x = torch.randn(b, c, h, w, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    oh, ow = h + pad[0] + pad[1] - 3, w + pad[0] + pad[1] - 3
    noise = torch.randn(noise_batch, 1, oh, ow, generator=g).to(DEV)
    bias, nw = torch.randn(c, generator=g).to(DEV), torch.tensor([0.41], device=DEV)
    gy = torch.randn(b, c, oh, ow, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    v = torch.randn(b, c, oh, ow, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        xs, bs, ns = x.clone().requires_grad_(True), bias.clone().requires_grad_(True), nw.clone().requires_grad_(True)
        gys = gy.clone().requires_grad_(True)
        if fused:
            y = blur_bias_act(xs, fir, pad, bs, noise, ns, 0.2, 1.3)
        else:
            y = fused_bias_noise_leaky_relu(ops.upfirdn2d(xs, fir, pad=pad), bs, noise, ns, 0.2, 1.3)
        gx, gb, gn = torch.autograd.grad(y, (xs, bs, ns), gys, create_graph=True)
        ggy, = torch.autograd.grad(gx, gys, x.clone(), retain_graph=True)           # second order, linear in gy
        res.append((y, gx, gb, gn, ggy))
    from oracle import ops as oo
    ref = oo.fused_leaky_relu(oo.upfirdn2d(x.float().cpu(), fir.cpu(), pad=pad) +
                              nw.cpu() * noise.cpu(), bias.cpu(), 0.2, 1.3)
    e_fused, e_two = rel_err(res[0][0].float(), ref), rel_err(res[1][0].float(), ref)
    assert e_fused < 1e-2 and e_fused <= e_two * 1.05 + 1e-6            # no less accurate than the two-pass form
    assert rel_err(res[0][0].float(), res[1][0].float()) < 1e-2
    # the gradient masks come from the sign of the outputs, which may differ where an output sits within rounding of
    # zero: a handful of elements change slope, so the gradients are compared in the L2 sense
    for a, r in zip(res[0][1:], res[1][1:]):
        assert (a.float() - r.float()).norm() / r.float().norm() < 2e-2
    assert v is not None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(1, 1, (2, 1)), (1, 2, (1, 1)), (2, 1, (2, 1)), (1, 1, (2, 2))])
def test_upfirdn2d_on_channel_slices(cfg, dtype):
    """A channel-slice of a wider channels-last map (the gradient of one piece of a concatenation) goes through
    msg_upfirdn2d_pitched: same result, bit for bit, as filtering the compacted copy."""
    ops = _ops()
    up, down, pad = cfg
    g = torch.Generator().manual_seed(up * 10 + down)
    fir = torch.randn(4, 4, generator=g).to(DEV)
    wide = torch.randn(2, 48, 14, 18, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    for lo, hi in ((0, 16), (16, 48), (8, 40)):
        piece = wide[:, lo:hi]
        assert not piece.is_contiguous(memory_format=torch.channels_last)
        y = ops.upfirdn2d(piece, fir, up=up, down=down, pad=pad)
        ref = ops.upfirdn2d(piece.contiguous(memory_format=torch.channels_last), fir, up=up, down=down, pad=pad)
        assert torch.equal(y, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 1, (2, 1)), (1, 1, (2, 1)), (1, 2, (1, 1))])
def test_upfirdn2d_into_channel_slice(cfg, dtype):
    """upfirdn2d(out=slice): the result written into its channel-slice of a wider channels-last map
    (msg_upfirdn2d_pitched2) equals the dense result bit for bit, leaves the other channels alone, and the gradient is the
    dense call's."""
    ops = _ops()
    up, down, pad = cfg
    g = torch.Generator().manual_seed(3)
    fir = torch.randn(4, 4, generator=g).to(DEV)
    x = torch.randn(2, 16, 12, 10, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    ref = ops.upfirdn2d(x, fir, up=up, down=down, pad=pad)
    oh, ow = ref.shape[2:]
    for lo, total in ((0, 16), (0, 48), (16, 48), (24, 40)):
        wide = torch.full((2, oh, ow, total), 7.0, device=DEV, dtype=dtype).permute(0, 3, 1, 2)
        xr = x.clone().requires_grad_(True)
        y = ops.upfirdn2d(xr, fir, up=up, down=down, pad=pad, out=wide[:, lo:lo + 16])
        assert y.data_ptr() == wide[:, lo:lo + 16].data_ptr() and torch.equal(y, ref)
        rest = torch.cat([wide[:, :lo], wide[:, lo + 16:]], dim=1)
        assert bool((rest == 7.0).all())
        gy = torch.randn(ref.shape, generator=g).to(DEV, dtype)
        gx, = torch.autograd.grad(y, xr, gy)
        xd = x.clone().requires_grad_(True)
        gd, = torch.autograd.grad(ops.upfirdn2d(xd, fir, up=up, down=down, pad=pad), xd, gy)
        assert torch.equal(gx, gd)
    from multi_stylegan_amd._lib import MsgHipError
    with pytest.raises(MsgHipError):                       # a destination that is not a channels-last slice
        ops.upfirdn2d(x, fir, up=up, down=down, pad=pad, out=torch.empty(2, 16, oh, ow, device=DEV, dtype=dtype))


def test_upfirdn2d_edge_cases():
    ops = _ops()
    from multi_stylegan_amd._lib import MsgHipError
    fir = torch.ones(4, 4, device=DEV) / 16
    with pytest.raises(MsgHipError):                       # CPU tensors: no silent fallback
        ops.upfirdn2d(torch.zeros(1, 1, 4, 4), fir.cpu())
    with pytest.raises(MsgHipError):                       # empty output
        ops.upfirdn2d(torch.zeros(1, 1, 2, 2, device=DEV), fir, pad=(0, 0))
    with pytest.raises(MsgHipError):                       # unsupported storage type
        ops.upfirdn2d(torch.zeros(1, 1, 8, 8, device=DEV, dtype=torch.int32), fir, pad=(2, 1))
    y64 = ops.upfirdn2d(torch.ones(1, 1, 8, 8, device=DEV, dtype=torch.float64), fir, pad=(2, 1))   # double: provided
    assert y64.dtype == torch.float64 and abs(float(y64[0, 0, 4, 4]) - 1.0) < 1e-15
    y = ops.upfirdn2d(torch.zeros(0, 8, 8, 8, device=DEV), fir, pad=(2, 1))      # empty batch
    assert y.shape == (0, 8, 8, 8)
    x = torch.ones(1, 8, 6, 6, device=DEV)
    up = ops.upfirdn2d(x, fir, up=2, pad=(2, 1))           # sum-1 FIR, no gain: interior == 1/4 (quirk Q3)
    assert torch.allclose(up[0, 0, 3:9, 3:9], torch.full((6, 6), 0.25, device=DEV))


def test_upfirdn2d_full_size_properties():
    """BASELINE config sizes (B=2 of the 16): linearity, adjointness <F x, g> == <x, F^T g>, and agreement between
    the NCHW (generic) and channels-last (vector) kernels on the same data."""
    ops = _ops()
    torch.manual_seed(0)
    fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64 * 4).to(DEV)
    x = torch.randn(2, 512, 256, 256, device=DEV).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn_like(x)
    y = ops.upfirdn2d(x, fir, pad=(2, 1))
    y2 = ops.upfirdn2d(x2, fir, pad=(2, 1))
    ysum = ops.upfirdn2d(x + 2 * x2, fir, pad=(2, 1))
    assert rel_err(ysum, y + 2 * y2) < 1e-5
    g = torch.randn_like(y)
    xg = x.detach().clone().requires_grad_(True)
    gx, = torch.autograd.grad(ops.upfirdn2d(xg, fir, pad=(2, 1)), xg, g)
    lhs, rhs = (y.double() * g.double()).sum(), (x.double() * gx.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-6
    sub = x[:1, :64].contiguous()                                            # NCHW planes -> generic kernel
    assert rel_err(ops.upfirdn2d(sub, fir, pad=(2, 1)), y[:1, :64]) < 1e-6
    # D-side up-2 and its adjoint (down-2) at 256 ch @ 128 -> 256
    fir1 = fir / 4
    u = torch.randn(2, 256, 128, 128, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    up = ops.upfirdn2d(u, fir1, up=2, pad=(2, 1))
    assert up.shape == (2, 256, 256, 256)
    gu = torch.randn_like(up)
    gin, = torch.autograd.grad(up, u, gu)
    lhs, rhs = (up.double() * gu.double()).sum(), (u.double() * gin.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-6


@pytest.mark.parametrize("case", ["mlp_2d", "conv_4d", "conv_4d_sqrt2"])
def test_fused_leaky_relu_golden(golden, case):
    z, ops = golden("fused_act"), _ops()
    x, b = z[case + ".x"].to(DEV).requires_grad_(True), z[case + ".b"].to(DEV).requires_grad_(True)
    gy = z[case + ".gy"].to(DEV).requires_grad_(True)
    y = ops.fused_leaky_relu(x, b, 0.2, float(z[case + ".scale"]))
    gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
    ggy, = torch.autograd.grad((gx, gb), gy, (z[case + ".ggx"].to(DEV), z[case + ".ggb"].to(DEV)))
    for got, key in ((y, "y"), (gx, "gx"), (gb, "gb"), (ggy, "ggy")):
        assert rel_err(got, z[f"{case}.{key}"]) < TOL32, key


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("shape,noise_batch", [((3, 16, 6, 5), 3), ((2, 24, 4, 4), 1), ((2, 40, 9, 7), 2),
                                               ((4, 8, 16, 16), 4), ((2, 6, 5, 5), 2)])
def test_fused_bias_noise_act_vs_oracle(shape, noise_batch, layout, dtype, tol):
    from oracle import ops as oo
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dtype).float()
    bias = torch.randn(shape[1], generator=g)
    noise = torch.randn(noise_batch, 1, *shape[2:], generator=g)
    nw = torch.randn(1, generator=g)
    gy = torch.randn(*shape, generator=g).to(dtype).float()
    ggx, ggb, ggw = torch.randn(*shape, generator=g).to(dtype).float(), torch.randn(shape[1], generator=g), \
        torch.randn(1, generator=g)
    xr, br, wr, gyr = [t.clone().requires_grad_(True) for t in (x, bias, nw, gy)]
    yr = oo.fused_leaky_relu(oo.noise_injection(xr, wr, noise), br, 0.2, 1.0)
    gr = torch.autograd.grad(yr, (xr, br, wr), gyr, create_graph=True)
    ggyr, = torch.autograd.grad(gr, gyr, (ggx, ggb, ggw))
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    mv = lambda t: t.to(DEV, dtype).contiguous(memory_format=fmt)
    xd, gyd = mv(x).requires_grad_(True), mv(gy).requires_grad_(True)
    bd, wd = bias.to(DEV).requires_grad_(True), nw.to(DEV).requires_grad_(True)
    y = ops.fused_bias_noise_leaky_relu(xd, bd, noise.to(DEV), wd, 0.2, 1.0)
    gd = torch.autograd.grad(y, (xd, bd, wd), gyd, create_graph=True)
    ggy, = torch.autograd.grad(gd, gyd, (mv(ggx), ggb.to(DEV), ggw.to(DEV)))
    assert rel_err(y.float(), yr) < tol
    assert rel_err(gd[0].float(), gr[0]) < tol
    assert rel_err(gd[1], gr[1]) < tol * (1 if dtype == torch.float32 else 4)
    assert rel_err(gd[2], gr[2]) < tol * (1 if dtype == torch.float32 else 4)
    assert rel_err(ggy.float(), ggyr) < tol


def test_fused_act_full_size_properties():
    """[4,512,256,256] channels-last bf16: mask consistency and grad_bias == column sums of grad_input."""
    ops = _ops()
    torch.manual_seed(1)
    x = torch.randn(4, 512, 256, 256, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(512, device=DEV, requires_grad=True)
    xg = x.clone().requires_grad_(True)
    y = ops.fused_leaky_relu(xg, b, 0.2, 1.0)
    ref = torch.nn.functional.leaky_relu(x.float() + b.detach()[None, :, None, None], 0.2)
    assert rel_err(y.float(), ref) < TOL16
    gy = torch.randn_like(y)
    gx, gb = torch.autograd.grad(y, (xg, b), gy)
    want_gx = gy.float() * torch.where(y.float() > 0, 1.0, 0.2)
    assert rel_err(gx.float(), want_gx) < TOL16
    assert rel_err(gb, gx.float().sum(dim=(0, 2, 3))) < 2e-3


def test_streaming_ops_above_two_gib():
    """The HBM-bound kernels on a map of more than 2^31 bytes / 2^30 elements ([33, 512, 256, 256] bf16, channels-last): the 4x4
    blur, the blur with the fused noise / bias / activation stage and the stand-alone activation, forward and backward, must agree
    with the same op on the first and on the LAST samples alone (the last sample starts beyond 2 GiB: 32-bit offsets anywhere in the
    path would drop or alias it)."""
    ops = _ops()
    b, c, r = 33, 512, 256
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(b, c, r, r, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)
    assert x.numel() * 2 > (1 << 31)
    fir = (torch.outer(torch.tensor([1., 3., 3., 1.]), torch.tensor([1., 3., 3., 1.])) / 64 * 4).to(DEV)
    bias = torch.randn(c, device=DEV, generator=g)
    noise = torch.randn(b, 1, r, r, device=DEV, generator=g)
    nw = torch.tensor([0.3], device=DEV)
    gy = torch.randn(b, c, r, r, device=DEV, dtype=torch.bfloat16, generator=g).contiguous(memory_format=torch.channels_last)

    def cl(t):
        return t.contiguous(memory_format=torch.channels_last)

    def run(op, xs, ns, gs):
        xs = xs.detach().requires_grad_(True)
        y = op(xs, ns)
        gx, = torch.autograd.grad(y, xs, gs)
        return y.detach(), gx.detach()

    cases = {
        "blur": lambda xs, ns: ops.upfirdn2d(xs, fir, pad=(2, 1)),
        "blur + noise / bias / activation": lambda xs, ns: ops.blur_bias_act(xs, fir, (2, 1), bias, ns, nw, scale=2 ** 0.5),
        "noise / bias / activation": lambda xs, ns: ops.fused_bias_noise_leaky_relu(xs, bias, ns, nw, scale=2 ** 0.5),
    }
    for name, op in cases.items():
        y, gx = run(op, x, noise, gy)
        for lo, hi in ((0, 2), (b - 2, b)):
            ys, gxs = run(op, cl(x[lo:hi]), noise[lo:hi].contiguous(), cl(gy[lo:hi]))
            assert torch.equal(y[lo:hi], ys), f"{name}: forward, samples {lo}..{hi}"
            assert torch.equal(gx[lo:hi], gxs), f"{name}: input gradient, samples {lo}..{hi}"
        del y, gx


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 17, 8), (2, 33, 1000), (2, 64, 1024), (1, 9, 4096), (5, 520)])
def test_softmax_rows(dtype, shape):
    """Attention softmax (u_net_2d_discriminator.py:378) against the CPU fp32 softmax on the same (rounded) input:
    value, gradient, and the second-order terms R1 needs."""
    from multi_stylegan_amd.op_static import softmax_rows
    torch.manual_seed(11)
    x_cpu = (3.0 * torch.randn(*shape)).to(dtype).float()
    gy_cpu = torch.randn(*shape).to(dtype).float()
    v_cpu = torch.randn(*shape).to(dtype).float()
    def run(x, gy, v, fn):
        x = x.clone().requires_grad_(True)
        gy = gy.clone().requires_grad_(True)
        y = fn(x)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        ggy, gx2 = torch.autograd.grad(gx, (gy, x), v)
        return [t.detach().float().cpu() for t in (y, gx, ggy, gx2)]
    ref = run(x_cpu, gy_cpu, v_cpu, lambda t: torch.softmax(t, dim=-1))
    got = run(x_cpu.to(DEV, dtype), gy_cpu.to(DEV, dtype), v_cpu.to(DEV, dtype), softmax_rows)
    tol = 2e-6 if dtype == torch.float32 else 1.5e-2
    for name, a, b in zip(("y", "gx", "d/dgy", "d/dx"), got, ref):
        assert rel_err(a, b) < tol, (name, rel_err(a, b))
    assert abs(got[0].sum(dim=-1) - 1).max() < (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gamma_merge(dtype):
    """(gamma * a + b) / sqrt(2) with gamma a 0-d parameter (NonLocalBlock merge, u_net_2d_discriminator.py:381) in one launch
    forward and one pass backward: against the torch formulation in fp64, bit-identical over repeated launches, and through
    a second-order pass."""
    ops = _ops()
    torch.manual_seed(4)
    a0 = torch.randn(3, 48, 20, 24, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    b0 = torch.randn_like(a0)
    gain = 1 / math.sqrt(2)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    outs = []
    for _ in range(2):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        gamma = torch.tensor(0.7, device=DEV, requires_grad=True)
        y = ops.gamma_merge(a, b, gamma, gain)
        gy = torch.randn(y.shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        torch.manual_seed(5)
        gy = torch.randn(y.shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        ga, gb, gg = torch.autograd.grad(y, (a, b, gamma), gy)
        outs.append((y.detach(), ga, gb, gg))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    ad, bd, gd = a0.double().requires_grad_(True), b0.double().requires_grad_(True), torch.tensor(0.7, device=DEV, dtype=torch.float64, requires_grad=True)
    yd = (gd * ad + bd) * gain
    want = torch.autograd.grad(yd, (ad, bd, gd), gy.double())
    y, ga, gb, gg = outs[0]
    assert rel_err(y.double(), yd) < tol and rel_err(ga.double(), want[0]) < tol and rel_err(gb.double(), want[1]) < tol
    assert abs(float(gg) - float(want[2])) < tol * max(1.0, abs(float(want[2])))
    assert gg.shape == () and gg.dtype == torch.float32
    # second order: d/d(gy) of <ga, ga> runs through the torch formulation of the backward
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    gamma = torch.tensor(0.7, device=DEV, requires_grad=True)
    gyr = gy.clone().requires_grad_(True)
    ga, = torch.autograd.grad(ops.gamma_merge(a, b, gamma, gain), a, gyr, create_graph=True)
    h, = torch.autograd.grad(ga.float().square().sum(), gyr)
    assert rel_err(h.double(), 2 * (0.7 * gain) ** 2 * gy.double()) < 5 * tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 6, 8, 8), (2, 6, 64, 64), (2, 3, 16, 32), (1, 6, 4, 4)])
@pytest.mark.parametrize("with_skip", [True, False])
def test_rgb_skip_merge_vs_oracle(dtype, shape, with_skip):
    """op_static.rgb_skip (csrc/rgb_skip.hip): conv.float() + bias + upfirdn2d(skip, up = 2, pad (2, 1)) -- OutputBlock.forward
    of the reference (multi_stylegan_generator.py:519-523) -- against the CPU oracle's upfirdn2d with an ASYMMETRIC 4 x 4 FIR
    (so that a flipped or transposed tap table fails): output, first-order gradients of all three operands, and the
    second-order pass (gradient of a function of the first-order gradients), which the path-length regulariser runs."""
    from multi_stylegan_amd import conv_ops
    from multi_stylegan_amd.op_static import rgb_skip
    from oracle import ops as oo
    b, c, h, w = shape
    gen = torch.Generator().manual_seed(h * 7 + c)
    fir = torch.rand(4, 4, generator=gen) + 0.1
    fir = fir / fir.sum()
    conv0 = torch.randn(b, c, h, w, generator=gen).to(dtype).float()          # (values the storage type holds exactly)
    bias0 = torch.randn(c, generator=gen)
    skip0 = torch.randn(b, c, h // 2, w // 2, generator=gen) if with_skip else None
    gy = torch.randn(b, c, h, w, generator=gen)
    probe = [torch.randn(b, c, h, w, generator=gen), torch.randn(c, generator=gen),
             torch.randn(b, c, h // 2, w // 2, generator=gen)]

    def run(dev, fused):
        conv = conv0.to(dev).requires_grad_(True)
        bias = bias0.to(dev).requires_grad_(True)
        skip = skip0.to(dev).requires_grad_(True) if with_skip else None
        leaves = [conv, bias] + ([skip] if with_skip else [])
        gyl = gy.to(dev).requires_grad_(True)
        if fused:
            cl = conv_ops.to_compute_layout(conv, dtype)
            assert rgb_skip.supported(cl, skip, fir.to(dev), 2, (2, 1))
            y = rgb_skip.rgb_skip_merge(cl, bias, skip, fir.to(dev))
        else:
            y = conv + bias.view(1, -1, 1, 1)
            if with_skip:
                y = y + oo.upfirdn2d(skip, fir.to(dev), up=2, down=1, pad=(2, 1))
        grads = torch.autograd.grad(y, leaves, gyl, create_graph=True)
        second = torch.autograd.grad(sum((g.float() * p.to(dev)).sum() for g, p in zip(grads, probe)), gyl)[0]
        return [y] + list(grads) + [second]

    got, want = run(DEV, True), run("cpu", False)
    names = ["y", "g_conv", "g_bias"] + (["g_skip"] if with_skip else []) + ["second"]
    for name, a, r in zip(names, got, want):
        # (g_conv is stored in the conv's dtype, and the second-order result is a function of it)
        tol = 1e-5 if (dtype == torch.float32 or name not in ("g_conv", "second")) else 1e-2
        assert a.shape == r.shape and rel_err(a.float(), r) < tol, (name, rel_err(a.float(), r))
    again = run(DEV, True)
    assert all(torch.equal(a, b2) for a, b2 in zip(got, again))                         # gathers only: bit-reproducible
