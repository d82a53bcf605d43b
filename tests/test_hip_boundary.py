"""Drop-in boundary proof: the two stub modules printed in INTEGRATION.md section 1 -- the files a
maintainer would put where the reference's `upfirdn2d_cuda` / `fused_act_cuda` CUDA extensions used to be -- are
extracted from that document, executed as written, and driven with the call patterns of the reference's own autograd
classes (multi_stylegan/op_static/upfirdn2d.py:22-145, fused_act.py:22-73: `[major, h, w, minor]` planes, empty tensors
for absent bias / refer, swapped up/down + flipped kernel + g_pad in backward, grad=1 with refer = saved output) against
the golden vectors captured from the reference."""
import os
import re
import types

import pytest
import torch

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# storage type -> tolerance relative to max|ref| (inputs are rounded to the storage type first; fp32 arithmetic)
TOLS = {torch.float32: 1e-5, torch.float16: 3e-3, torch.bfloat16: 2e-2, torch.float64: 1e-12}


def _want(z, key, dtype, fn):
    """The expected tensor: the reference-generated golden vector, except in float64, where the fixtures (float32 data) are
    too coarse for a 1e-12 check -- there the pinned oracle (oracle/ops.py, checked against the same fixtures in
    tests/test_oracle_golden.py) is evaluated in float64 on the same inputs."""
    return fn() if dtype == torch.float64 else z[key]
UPFIRDN_CASES = ["g_blur_pad21_gain4", "g_skip_up2_pad21", "d_blur_pad22_odd", "bwd_of_up2_down2",
                 "asym_blur_pad21", "asym_up2_pad21", "asym_down2_pad12", "blur_pad11"]


@pytest.fixture(scope="module")
def stubs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    out = {}
    for name in ("upfirdn2d_cuda", "fused_act_cuda"):
        src = next(b for b in blocks if b.startswith(f"# {name}.py"))
        src = src.replace('"libmsg_hip.so"', repr(os.path.join(ROOT, "multi_stylegan_amd", "libmsg_hip.so")))
        mod = types.ModuleType(name)
        exec(compile(src, f"INTEGRATION.md:{name}", "exec"), mod.__dict__)
        out[name] = mod
    return out


def _planes(t):
    """[B,C,H,W] -> the reference's [major = B*C, H, W, minor = 1] (upfirdn2d.py:98-99)."""
    return t.reshape(-1, t.shape[2], t.shape[3], 1)


@pytest.mark.parametrize("dtype", list(TOLS))
@pytest.mark.parametrize("case", UPFIRDN_CASES)
def test_upfirdn2d_stub_module(golden, stubs, case, dtype):
    z, up_mod, tol = golden("upfirdn2d"), stubs["upfirdn2d_cuda"], TOLS[dtype]
    up, down, p0, p1 = [int(v) for v in z[case + ".cfg"]]
    fir = z[case + ".fir"].to(DEV)
    kh, kw = fir.shape
    x = z[case + ".x"].to(DEV, dtype)
    b, c, in_h, in_w = x.shape
    y = up_mod.upfirdn2d(_planes(x), fir, up, up, down, down, p0, p1, p0, p1)
    out_h, out_w = y.shape[1:3]
    assert y.dtype == dtype and y.shape == (b * c, out_h, out_w, 1)
    from oracle import ops as oo
    f64 = lambda t: t.double().cpu()
    want_y = _want(z, case + ".y", dtype, lambda: oo.upfirdn2d(f64(x), f64(fir), up, down, (p0, p1)))
    assert rel_err(y.reshape(b, c, out_h, out_w), want_y) < tol
    # backward = the same op with up <-> down, the flipped kernel and g_pad (upfirdn2d.py:34-45, 114-119)
    g0x, g1x = kw - p0 - 1, in_w * up - out_w * down + p0 - up + 1
    g0y, g1y = kh - p0 - 1, in_h * up - out_h * down + p0 - up + 1
    gy = z[case + ".gy"].to(DEV, dtype)
    gx = up_mod.upfirdn2d(_planes(gy), torch.flip(fir, [0, 1]), down, down, up, up, g0x, g1x, g0y, g1y)
    def adjoint():                                  # d<oracle(x), gy>/dx in float64
        x64 = f64(x).requires_grad_(True)
        return torch.autograd.grad(oo.upfirdn2d(x64, f64(fir), up, down, (p0, p1)), x64, f64(gy))[0]
    assert rel_err(gx.reshape(b, c, in_h, in_w), _want(z, case + ".gx", dtype, adjoint)) < tol
    # double backward = the forward configuration on the incoming second-order gradient (upfirdn2d.py:71-82)
    ggx = z[case + ".ggx"].to(DEV, dtype)
    ggy = up_mod.upfirdn2d(_planes(ggx), fir, up, up, down, down, p0, p1, p0, p1)
    want_ggy = _want(z, case + ".ggy", dtype, lambda: oo.upfirdn2d(f64(ggx), f64(fir), up, down, (p0, p1)))
    assert rel_err(ggy.reshape(b, c, out_h, out_w), want_ggy) < tol


@pytest.mark.parametrize("dtype", list(TOLS))
@pytest.mark.parametrize("case", ["mlp_2d", "conv_4d", "conv_4d_sqrt2"])
def test_fused_bias_act_stub_module(golden, stubs, case, dtype):
    z, act_mod, tol = golden("fused_act"), stubs["fused_act_cuda"], TOLS[dtype]
    scale = float(z[case + ".scale"])
    x, bias = z[case + ".x"].to(DEV, dtype), z[case + ".b"].to(DEV)
    empty = torch.empty(0, device=DEV)                                   # "absent" (fused_act.py:27, 59)
    out = act_mod.fused_bias_act(x, bias, empty, 3, 0, 0.2, scale)
    from oracle import ops as oo
    f64 = lambda t: t.double().cpu()
    bshape = (1, -1) + (1,) * (x.ndim - 2)
    # (the native signature takes `float alpha, float scale` -- fused_bias_act.cpp:11-12 -- so the float64 arithmetic sees the
    #  float32-rounded 0.2 and sqrt(2), in the reference as here)
    a32, s32 = float(torch.tensor(0.2, dtype=torch.float32)), float(torch.tensor(scale, dtype=torch.float32))
    want_y = _want(z, case + ".y", dtype, lambda: oo.fused_leaky_relu(f64(x), f64(bias), a32, s32))
    assert out.dtype == dtype and rel_err(out, want_y) < tol
    # backward: grad=1, slope from the sign of the saved OUTPUT, no bias (fused_act.py:31-33); grad_bias in PyTorch
    gy = z[case + ".gy"].to(DEV, dtype)
    ref_out = out if dtype == torch.float64 else z[case + ".y"].to(DEV, dtype)
    gx = act_mod.fused_bias_act(gy, empty, ref_out, 3, 1, 0.2, scale)
    one, slope = torch.tensor(1.0, dtype=torch.float64), torch.tensor(a32, dtype=torch.float64)
    mask = lambda: torch.where(f64(ref_out) > 0, one, slope) * s32
    assert rel_err(gx, _want(z, case + ".gx", dtype, lambda: f64(gy) * mask())) < tol
    dims = [0] + list(range(2, gx.ndim))
    want_gb = _want(z, case + ".gb", dtype, lambda: (f64(gy) * mask()).sum(dims))
    assert rel_err((gx if dtype == torch.float64 else gx.float()).sum(dims), want_gb) < max(tol, 1e-5) * 4
    # double backward: grad=1 on (gg_input + gg_bias[c]) with the same refer (fused_act.py:47-49)
    ggx, ggb = z[case + ".ggx"].to(DEV, dtype), z[case + ".ggb"].to(DEV)
    ggy = act_mod.fused_bias_act(ggx, ggb, ref_out, 3, 1, 0.2, scale)
    want_ggy = _want(z, case + ".ggy", dtype, lambda: (f64(ggx) + f64(ggb).view(bshape)) * mask())
    assert rel_err(ggy, want_ggy) < tol


def test_stub_modules_error_behaviour(stubs):
    """CHECK_CUDA of the reference (upfirdn2d.cpp:8,15-16, fused_bias_act.cpp:7,13-14) is a TORCH_CHECK: host tensors raise
    RuntimeError, exactly what a caller of the CUDA modules catches; a dtype outside the dispatch list is refused loudly."""
    up_mod, act_mod = stubs["upfirdn2d_cuda"], stubs["fused_act_cuda"]
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        up_mod.upfirdn2d(torch.zeros(1, 4, 4, 1), torch.ones(4, 4), 1, 1, 1, 1, 2, 1, 2, 1)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        act_mod.fused_bias_act(torch.zeros(2, 3), torch.zeros(3), torch.empty(0), 3, 0, 0.2, 1.0)
    with pytest.raises(KeyError):
        up_mod.upfirdn2d(torch.zeros(1, 4, 4, 1, device=DEV, dtype=torch.int32), torch.ones(4, 4, device=DEV),
                         1, 1, 1, 1, 2, 1, 2, 1)


def test_gradcheck_runs_in_float64_through_the_drop_in_ops():
    """What SURVEY 8c did with CPU stand-ins, on the device through the C ABI: gradcheck and gradgradcheck of the
    twice-differentiable drop-in ops (`op_static.upfirdn2d`, `fused_leaky_relu`: the reference's public names and argument
    lists, op_static/upfirdn2d.py:148-153, fused_act.py:88-89) in float64, i.e. through msg_upfirdn2d / msg_fused_bias_act
    with MSG_F64 -- the same entries the stub modules above bind."""
    from multi_stylegan_amd import op_static
    torch.manual_seed(11)
    fir = torch.tensor([1., 3., 3., 1.], dtype=torch.float64, device=DEV)
    fir = torch.outer(fir, fir) / 64
    for up, down, p0, p1 in ((1, 1, 2, 1), (2, 1, 2, 1), (1, 2, 1, 1)):
        x = torch.randn(2, 3, 6, 5, dtype=torch.float64, device=DEV, requires_grad=True)
        own = lambda t: op_static.upfirdn2d(t, fir, up=up, down=down, pad=(p0, p1))
        assert torch.autograd.gradcheck(own, (x,), eps=1e-6, atol=1e-8)
        assert torch.autograd.gradgradcheck(own, (x,), eps=1e-6, atol=1e-8)
        xcl = torch.randn(2, 3, 6, 5, dtype=torch.float64, device=DEV).contiguous(memory_format=torch.channels_last)
        assert torch.autograd.gradcheck(own, (xcl.requires_grad_(True),), eps=1e-6, atol=1e-8)
    x = torch.randn(3, 4, 5, 5, dtype=torch.float64, device=DEV)
    x = (x + 0.3 * x.sign()).requires_grad_(True)                       # keep clear of the kink at 0
    bias = (0.05 * torch.randn(4, dtype=torch.float64, device=DEV)).requires_grad_(True)
    act = lambda t, b: op_static.fused_leaky_relu(t, b, 0.2, 2 ** 0.5)
    assert torch.autograd.gradcheck(act, (x, bias), eps=1e-6, atol=1e-8)
    assert torch.autograd.gradgradcheck(act, (x, bias), eps=1e-6, atol=1e-8)


def test_product_ops_accept_half():
    """The drop-in Python ops (`op_static.upfirdn2d`, `fused_leaky_relu`) in float16 storage, as the reference's
    `half` dispatch allows: forward and gradient against the fp32 result on the same rounded inputs."""
    from multi_stylegan_amd import op_static
    torch.manual_seed(4)
    x = torch.randn(2, 16, 12, 12, device=DEV).half()
    fir = torch.tensor([1., 3., 3., 1.], device=DEV)
    fir = torch.outer(fir, fir) / 64
    for up, down, pad in ((1, 1, (2, 1)), (2, 1, (2, 1)), (1, 2, (1, 1))):
        xh = x.clone().requires_grad_(True)
        xf = x.float().requires_grad_(True)
        yh = op_static.upfirdn2d(xh, fir, up=up, down=down, pad=pad)
        yf = op_static.upfirdn2d(xf, fir, up=up, down=down, pad=pad)
        assert yh.dtype == torch.float16 and rel_err(yh, yf) < 2e-3
        g = torch.randn_like(yf)
        yh.backward(g.half()); yf.backward(g.half().float())
        assert rel_err(xh.grad, xf.grad) < 2e-3
    b = torch.randn(16, device=DEV, requires_grad=True)
    xh = x.clone().requires_grad_(True)
    xf = x.float().requires_grad_(True)
    yh = op_static.fused_leaky_relu(xh, b, 0.2, 1.0)
    yf = op_static.fused_leaky_relu(xf, b, 0.2, 1.0)
    assert yh.dtype == torch.float16 and rel_err(yh, yf) < 2e-3
