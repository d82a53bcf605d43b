"""``upfirdn2d`` as a twice-differentiable autograd op on the gfx950 kernel.

Mirrors the public interface of the reference's op_static/upfirdn2d.py:148-153.  The backward of an
(up, down, pad) FIR pass is another FIR pass with up and down swapped, the FIR flipped and the padding of
upfirdn2d.py:114-119; the backward of that is the original configuration again (upfirdn2d.py:66-88), so one
native entry point serves all three orders.

Memory layout: a channels-last input ([B,C,H,W] with NHWC strides) is handed to the kernel as
``major = B, minor = C``, anything else as NCHW planes ``major = B*C, minor = 1`` -- the two layouts the
reference's native signature already distinguishes (upfirdn2d.cpp:12-19).
"""

import torch
from torch.autograd import Function

from .. import _lib
from .._autograd import _derive


def _is_channels_last(x):
    return x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _out_size(n, up, down, p0, p1, k):
    return (n * up + p0 + p1 - k) // down + 1


# 1: the bf16 blur takes the separable kernel (4.4 vs 2.8 TB/s); fp32 stays on the 2-D kernel, which is faster there
# (4.1 vs 3.5 TB/s: half the taps per byte).  0: always the 2-D kernels, 2: separable for both types (tests, A/B).
_SEPARABLE = 1      # which maps take the separable blur kernel: 1 = bf16 (fp32 stays on the 2-D kernel, faster there); tests set 0 / 2


def _separable(fir: torch.Tensor):
    """(fir_y, fir_x) on the device with fir == outer(fir_y, fir_x) to 1e-6 relative, or None.  Needs the FIR's values
    on the host, i.e. one synchronising copy -- done once per FIR tensor (module buffer) and cached on it."""
    hit = fir.__dict__.get("_msg_separable")
    if hit is not None and hit[0] == fir._version:
        return hit[1]
    k = fir.detach().to("cpu", torch.float64)
    i, j = divmod(int(k.abs().argmax()), k.shape[1])
    out = None
    if k[i, j] != 0:
        fy, fx = k[:, j] / k[i, j], k[i, :]
        if (torch.outer(fy, fx) - k).abs().max() <= 1e-6 * k.abs().max():
            out = (fy.to(fir.device, torch.float32).contiguous(), fx.to(fir.device, torch.float32).contiguous())
    fir.__dict__["_msg_separable"] = (fir._version, out)
    return out


_DT_NAME = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "f16"}


def _slice_pitch(t, c, h, w):
    """Pixel pitch (elements) of a [B,c,h,w] channels-last map or channel-slice of one that the pitched entry points
    take as it is, else None."""
    if t.shape[1:] != (c, h, w) or not t.is_cuda:
        return None
    sb, sc, sh, sw = t.stride()
    vec = 16 // t.element_size()
    ok = (sc == 1 or c == 1) and sw >= c and sh == w * sw and (sb == h * w * sw or t.shape[0] == 1) and sw % vec == 0 and \
        c % vec == 0 and t.data_ptr() % 16 == 0
    return sw if ok else None


def _launch(x, fir, up, down, pad, out=None):
    """x [B,C,H,W] (either layout), fir [kh,kw] fp32 on the same device -> y, same layout and dtype as x.
    out: a channels-last map or channel-slice of one that receives the result (see upfirdn2d)."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    dev = _lib.require_gpu(x, fir)
    b, c, h, w = x.shape
    kh, kw = fir.shape
    oh, ow = _out_size(h, up_y, down_y, py0, py1, kh), _out_size(w, up_x, down_x, px0, px1, kw)
    if oh <= 0 or ow <= 0:
        raise _lib.MsgHipError(f"upfirdn2d: empty output {oh}x{ow}")
    if x.dtype == torch.float64:
        # the `double` of the reference's dispatch (upfirdn2d_kernel.cu:225): storage, FIR and arithmetic in float64 -- the
        # precision gradcheck / gradgradcheck need.  One scalar kernel, either layout.
        cl = _is_channels_last(x)
        x = x if cl else x.contiguous()
        major, minor = (b, c) if cl else (b * c, 1)
        y = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        fir64 = fir.to(torch.float64).contiguous()
        with _lib.on_device(dev):
            code = _lib.lib().msg_upfirdn2d(x.data_ptr(), fir64.data_ptr(), y.data_ptr(), _lib.MSG_F64, major, h, w, minor,
                                            kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1, _lib.stream_of(dev))
        _lib.check(code, "msg_upfirdn2d")
        return y
    fir = fir.to(torch.float32).contiguous()
    pitch = None
    if out is not None:
        # the result goes straight into its slice of a wider channels-last map (no concatenation copy afterwards)
        out_pitch = _slice_pitch(out, c, oh, ow) if out.dtype == x.dtype and out.shape[0] == b else None
        in_pitch = _slice_pitch(x, c, h, w)
        if out_pitch is None or in_pitch is None or kh > 4 or kw > 4 or \
                x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise _lib.MsgHipError(f"upfirdn2d(out=): input {tuple(x.shape)} / {x.stride()} or destination "
                                   f"{tuple(out.shape)} / {out.stride()} is not a channels-last map or channel-slice")
        key = ("upfirdn2d", _DT_NAME.get(x.dtype, x.dtype), ("up", up_x, "down", down_x), "vec")
        with _lib.on_device(dev), _lib.kernel_clock.span(key, (b * c * h * w + b * c * oh * ow) * x.element_size()):
            code = _lib.lib().msg_upfirdn2d_pitched2(x.data_ptr(), fir.data_ptr(), out.data_ptr(), _lib.dtype_code(x, True),
                                                     b, h, w, c, in_pitch, out_pitch, kh, kw, up_x, up_y, down_x, down_y,
                                                     px0, px1, py0, py1, _lib.stream_of(dev))
        _lib.check(code, "msg_upfirdn2d_pitched2")
        return out
    if c > 1 and x.stride(1) == 1 and not _is_channels_last(x):
        # a channel-slice of a channels-last buffer (e.g. the gradient of one piece of a concatenation): filtered in
        # place through the pitched entry point when its pitch allows, else compacted in the SAME layout (one copy)
        sb, _, sh, sw = x.stride()
        vec = 16 // x.element_size()
        if sh == w * sw and (sb == h * w * sw or b == 1) and sw % vec == 0 and c % vec == 0 and x.data_ptr() % 16 == 0 \
                and kh <= 4 and kw <= 4 and x.dtype in (torch.float32, torch.bfloat16, torch.float16):
            pitch = sw
        else:
            x = x.contiguous(memory_format=torch.channels_last)
    if pitch is not None:
        major, minor = b, c
        y = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        key = ("upfirdn2d", _DT_NAME.get(x.dtype, x.dtype), ("up", up_x, "down", down_x), "vec")
        with _lib.on_device(dev), _lib.kernel_clock.span(key, (b * c * h * w + y.numel()) * x.element_size()):
            code = _lib.lib().msg_upfirdn2d_pitched(x.data_ptr(), fir.data_ptr(), y.data_ptr(), _lib.dtype_code(x, True),
                                                    major, h, w, minor, pitch, kh, kw, up_x, up_y, down_x, down_y,
                                                    px0, px1, py0, py1, _lib.stream_of(dev))
        _lib.check(code, "msg_upfirdn2d_pitched")
        return y
    if _is_channels_last(x):
        major, minor = b, c
        y = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    else:
        x = x.contiguous()
        major, minor = b * c, 1
        y = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev)
    vec_ok = minor % (16 // x.element_size()) == 0 and kh <= 4 and kw <= 4
    key = ("upfirdn2d", _DT_NAME.get(x.dtype, x.dtype), ("up", up_x, "down", down_x))       # (joined by the clock, when it is on)
    if vec_ok and minor > 1 and up == (1, 1) and down == (1, 1) and kh == 4 and kw == 4 and \
            (_SEPARABLE == 2 or (_SEPARABLE == 1 and x.dtype == torch.bfloat16)):
        factors = _separable(fir)
        if factors is not None:                       # the blur: separable sliding-window kernel (csrc/blur_sep.hip)
            with _lib.on_device(dev), _lib.kernel_clock.span(key + ("sep",), (x.numel() + y.numel()) * x.element_size()):
                code = _lib.lib().msg_upfirdn2d_separable(
                    x.data_ptr(), factors[0].data_ptr(), factors[1].data_ptr(), y.data_ptr(), _lib.dtype_code(x),
                    major, h, w, minor, kh, kw, px0, px1, py0, py1, _lib.stream_of(dev))
            _lib.check(code, "msg_upfirdn2d_separable")
            return y
    key += ("vec" if vec_ok else "generic",)
    with _lib.on_device(dev), _lib.kernel_clock.span(key, (x.numel() + y.numel()) * x.element_size()):
        code = _lib.lib().msg_upfirdn2d(x.data_ptr(), fir.data_ptr(), y.data_ptr(), _lib.dtype_code(x, True),
                                        major, h, w, minor, kh, kw, up_x, up_y, down_x, down_y,
                                        px0, px1, py0, py1, _lib.stream_of(dev))
    _lib.check(code, "msg_upfirdn2d")
    return y


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, fir, fir_flipped, up, down, pad, g_pad, in_hw):
        ctx.save_for_backward(fir)
        ctx.cfg = (up, down, pad)
        gin = _launch(grad_output, fir_flipped, down, up, g_pad)       # up <-> down swapped
        assert gin.shape[2:] == tuple(in_hw), (gin.shape, in_hw)
        return gin

    @staticmethod
    def backward(ctx, gradgrad_input):
        fir, = ctx.saved_tensors
        up, down, pad = ctx.cfg
        return UpFirDn2d.apply(gradgrad_input, fir, up, down, pad), None, None, None, None, None, None, None


def _flipped(fir: torch.Tensor) -> torch.Tensor:
    """The adjoint pass's FIR.  Cached on the (module buffer) tensor object: the flip is a 16-element kernel launched
    ~70 times per training step otherwise."""
    hit = fir.__dict__.get("_msg_flipped")
    if hit is not None and hit[0] == fir._version and hit[1].device == fir.device:
        return hit[1]
    with torch.no_grad():
        out = torch.flip(fir, [0, 1])
    fir.__dict__["_msg_flipped"] = (fir._version, out)
    return out


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, x, fir, up, down, pad, out=None):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = fir.shape
        h, w = x.shape[2:]
        y = _launch(x, fir, up, down, pad, out=out)
        if out is not None:
            y = out.view_as(out)                   # (a fresh alias: `out` itself is an input of this node)
        oh, ow = y.shape[2:]
        # padding of the adjoint pass (reference op_static/upfirdn2d.py:114-119)
        ctx.g_pad = (kw - px0 - 1, w * up_x - ow * down_x + px0 - up_x + 1,
                     kh - py0 - 1, h * up_y - oh * down_y + py0 - up_y + 1)
        ctx.cfg = (up, down, pad, (h, w))
        ctx.save_for_backward(fir, _flipped(fir))
        return y

    @staticmethod
    def backward(ctx, grad_output):
        fir, fir_flipped = ctx.saved_tensors
        up, down, pad, in_hw = ctx.cfg
        gin = _derive(UpFirDn2dBackward, grad_output, fir, fir_flipped, up, down, pad, ctx.g_pad, in_hw)
        return gin, None, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0), out=None):
    """FIR resampling of a [B,C,H,W] tensor; same arguments as the reference wrapper.  ``out`` (not in the reference): a
    channels-last map, or a channel-slice of a wider one, that receives the result -- the piece of a channel
    concatenation written in place; the returned tensor aliases it."""
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]), out)


def _blur_act_eligible(x, fir, pad):
    """The fused blur + activation kernel takes bf16 (or fp32 when _SEPARABLE == 2) channels-last maps, whole 16-byte
    channel vectors and a separable 4x4 FIR."""
    if not (x.is_cuda and x.ndim == 4 and _is_channels_last(x) and tuple(fir.shape) == (4, 4)):
        return False
    if x.shape[1] % (16 // x.element_size()) or not (_SEPARABLE == 2 or (_SEPARABLE == 1 and x.dtype == torch.bfloat16)):
        return False
    return _separable(fir.to(torch.float32).contiguous() if fir.dtype != torch.float32 else fir) is not None


class BlurBiasAct(Function):
    """blur (up = down = 1) -> (noise +) bias -> leaky ReLU in one launch (csrc/blur_sep.hip), for the upsampling
    StyledConv2d whose activation sits behind its blur.  Equal to upfirdn2d followed by the fused activation up to the
    intermediate rounding the two-pass form applies to the blur result (the fused kernel keeps it in fp32);
    the backward is composed of the same differentiable pieces (activation backward from the output's sign, then the
    adjoint FIR pass), so second-order terms are those of the two-pass form."""

    @staticmethod
    def forward(ctx, x, fir, pad, bias, noise, noise_w, alpha, scale, act_handle=None):
        px0, px1, py0, py1 = pad
        dev = _lib.require_gpu(x, fir, bias, noise, noise_w)
        b, c, h, w = x.shape
        oh, ow = _out_size(h, 1, 1, py0, py1, 4), _out_size(w, 1, 1, px0, px1, 4)
        fy, fx = _separable(fir)
        y = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
        nz = nw = None
        if noise is not None:
            if noise.shape[0] not in (1, b) or noise.shape[1] != 1 or tuple(noise.shape[2:]) != (oh, ow):
                raise _lib.MsgHipError(f"noise shape {tuple(noise.shape)} does not match output {(b, c, oh, ow)}")
            nz, nw = noise.detach().to(torch.float32).contiguous(), noise_w.detach().to(torch.float32).contiguous()
        key = ("upfirdn2d", "bf16" if x.dtype == torch.bfloat16 else "f32", "up1down1", "sep+act")    # blur + activation: own key
        nbytes = (x.numel() + y.numel()) * x.element_size() + (0 if nz is None else nz.numel() * 4)
        from .fused_act import sign_mask_for
        mask = sign_mask_for(b, c, oh, ow, x.dtype, dev) if any(ctx.needs_input_grad) else None   # (no backward: no bytes)
        with _lib.on_device(dev), _lib.kernel_clock.span(key, nbytes):
            code = _lib.lib().msg_upfirdn2d_separable_act_mask(
                x.data_ptr(), fy.data_ptr(), fx.data_ptr(), y.data_ptr(), _lib.dtype_code(x), b, h, w, c, 4, 4,
                px0, px1, py0, py1, _lib.ptr(b32), _lib.ptr(nz), _lib.ptr(nw), 1 if nz is None else nz.shape[0],
                float(alpha), float(scale), _lib.ptr(mask), _lib.stream_of(dev))
        _lib.check(code, "msg_upfirdn2d_separable_act_mask")
        ctx.mask = None if mask is None else (mask, 1, c)      # (plain [pixel][c / 8] order)
        ctx.g_pad = (4 - px0 - 1, w - ow + px0, 4 - py0 - 1, h - oh + py0)
        ctx.cfg = (pad, (h, w), float(alpha), float(scale), bias is not None, noise is not None,
                   None if noise_w is None else noise_w.shape)
        ctx.save_for_backward(fir, _flipped(fir), y, noise)
        # act_handle (conv_ops.ActHandle): the output's only consumer is a 3x3 conv that may run this activation's backward in
        # the epilogue of its data-gradient launch
        ctx.act_handle = None
        if act_handle is not None and mask is not None:
            act_handle.arm(y, ctx.mask, alpha, scale, None, bias is not None, noise)
            ctx.act_handle = act_handle if act_handle.armed else None
        return y

    @staticmethod
    def backward(ctx, gy):
        from .fused_act import FusedLeakyReLUFunctionBackward
        fir, fir_flipped, y, noise = ctx.saved_tensors
        pad, in_hw, alpha, scale, has_bias, has_noise, nw_shape = ctx.cfg
        handed = ctx.act_handle.done if ctx.act_handle is not None else None
        if handed is not None:                 # the consumer's data-gradient launch applied the activation's backward: gy IS gpre
            ctx.act_handle.done = None
            gpre, (gb, gnw) = gy, handed
        else:
            gpre, gb, gnw = _derive(FusedLeakyReLUFunctionBackward, gy, y, noise if has_noise else None, has_bias, alpha, scale,
                                    ctx.mask)
        gin = _derive(UpFirDn2dBackward, gpre, fir, fir_flipped, (1, 1), (1, 1), pad, ctx.g_pad, in_hw) \
            if ctx.needs_input_grad[0] else None
        return gin, None, None, (gb if has_bias and ctx.needs_input_grad[3] else None), None, \
            (gnw.reshape(nw_shape) if has_noise and ctx.needs_input_grad[5] else None), None, None, None


def blur_bias_act(input, kernel, pad, bias, noise, noise_weight, negative_slope=0.2, scale=1.0, act_handle=None):
    """leaky_relu(upfirdn2d(input, kernel, pad=pad) + noise_weight * noise + bias) * scale; one launch when eligible.
    act_handle: see BlurBiasAct.forward."""
    if _blur_act_eligible(input, kernel, pad):
        return BlurBiasAct.apply(input, kernel, (pad[0], pad[1], pad[0], pad[1]), bias, noise, noise_weight,
                                 float(negative_slope), float(scale), act_handle)
    from .fused_act import fused_bias_noise_leaky_relu
    return fused_bias_noise_leaky_relu(upfirdn2d(input, kernel, pad=pad), bias, noise, noise_weight, negative_slope, scale)
