"""Dense contractions of the hot path on the hand-written MFMA kernels (csrc/conv_fprop.hip, conv_wgrad.hip).

Three primitives per convolution geometry, closed under differentiation so that first AND second order autograd
(R1 on the discriminator, path-length regularisation on the generator) run on the same two kernels:

    F(x, w)  -> y        forward contraction
    D(gy, w) -> gx       data gradient        (same kernel as F with the weights re-laid)
    G(gy, x) -> gw       weight gradient      (TN kernel)

    dF = (D(gy, w), G(gy, x));   dD = (F(v, w), G(gy, v));   dG = (F(x, u), D(gy, u))

Geometries: "conv" (kh x kw, stride 1 or 2, any padding) and "up2" (the generator's 2x2 stride-2 transposed conv,
run as a 1x1 contraction to 4*O channels stored pixel-shuffled).  Weights are given in the reference's parameter
layout ([O,I,kh,kw], or [B,O,I,kh,kw] for the per-sample weights of the modulated convolution) and re-laid per call
into the kernels' K-contiguous form; activations are channels-last with a 16-byte-aligned channel stride.
There is no CPU or library fallback.
"""
import math
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib


# ----------------------------------------------------------------------------------------------- layout helpers
def _vec(dtype) -> int:
    return 8 if dtype == torch.bfloat16 else 4


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def to_compute_layout(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """Feature maps are kept channels-last (NHWC in HBM, NCHW logical shape)."""
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    if x.ndim == 4 and x.shape[1] > 1:
        return x.contiguous(memory_format=torch.channels_last)
    return x.contiguous()


def _nhwc_view(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """-> (tensor whose memory is [B, H, W, Cx] with the real channels first, Cx).  Accepts channels-last tensors and
    channel-slices of channels-last buffers as they are; anything else is copied into a zero-padded NHWC buffer."""
    b, c, h, w = x.shape
    vec = _vec(x.dtype)
    sb, sc, sh, sw = x.stride()
    cx = sw
    if (sc == 1 or c == 1) and cx >= c and cx % vec == 0 and sh == w * cx and (sb == h * w * cx or b == 1) \
            and x.data_ptr() % 16 == 0:
        return x, cx
    cx = _round_up(c, vec)
    buf = torch.zeros((b, cx, h, w), dtype=x.dtype, device=x.device).contiguous(memory_format=torch.channels_last) \
        if cx > 1 else torch.zeros((b, cx, h, w), dtype=x.dtype, device=x.device)
    if cx == c:
        buf.copy_(x)
        return buf, cx
    buf[:, :c].copy_(x)
    return buf[:, :c], cx


def _alloc_out(b: int, n: int, h: int, w: int, dtype, device) -> Tuple[torch.Tensor, int]:
    """Channels-last output with the channel stride rounded up to 16 bytes; returns (logical [b,n,h,w] view, ldy)."""
    ld = _round_up(n, _vec(dtype))
    buf = torch.empty((b, h, w, ld), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :n], ld


def _relay_fwd(w: torch.Tensor, dtype) -> Tuple[torch.Tensor, int]:
    """[..., O, I, kh, kw] -> [..., O, kh*kw, Ck] (K-contiguous, input channels zero-padded to a 128-byte run)."""
    *lead, o, i, kh, kw = w.shape
    ck = _round_up(i, 128 // (2 if dtype == torch.bfloat16 else 4))
    out = torch.zeros((*lead, o, kh * kw, ck), dtype=dtype, device=w.device)
    out[..., :i] = w.reshape(*lead, o, i, kh * kw).transpose(-1, -2)
    return out, ck


def _relay_dgrad(w: torch.Tensor, dtype, flip: bool) -> Tuple[torch.Tensor, int]:
    """[..., O, I, kh, kw] -> [..., I, kh*kw (flipped if asked), Ok]: the weights of the data-gradient contraction."""
    *lead, o, i, kh, kw = w.shape
    ok = _round_up(o, 128 // (2 if dtype == torch.bfloat16 else 4))
    src = w.reshape(*lead, o, i, kh * kw)
    if flip:
        src = src.flip(-1)
    out = torch.zeros((*lead, i, kh * kw, ok), dtype=dtype, device=w.device)
    out[..., :o] = src.permute(*range(len(lead)), len(lead) + 1, len(lead) + 2, len(lead))
    return out, ok


# --------------------------------------------------------------------------------------------------- raw launches
def _launch_fprop(x, wk, ck, bias, n, out_hw, kh, kw, stride, pad, in_up, pixel_shuffle, per_sample, c_real):
    dev = _lib.require_gpu(x, wk, bias)
    xv, cx = _nhwc_view(x)
    b, _, ih, iw = xv.shape
    oh, ow = out_hw
    if pixel_shuffle:
        y, ldy = _alloc_out(b, n // 4, 2 * oh, 2 * ow, x.dtype, dev)
    else:
        y, ldy = _alloc_out(b, n, oh, ow, x.dtype, dev)
    wstride = wk.stride(0) if per_sample else 0
    # algorithmic FLOPs: real channels, and only the taps a transposed strided conv can reach (1/in_up^2)
    flops = 2.0 * b * oh * ow * n * kh * kw * c_real / (in_up * in_up)
    with torch.cuda.device(dev), _lib.kernel_clock.span(f"conv_fprop/{'bf16' if x.dtype == torch.bfloat16 else 'f32'}", flops):
        code = _lib.lib().msg_conv2d_fprop(
            xv.data_ptr(), wk.data_ptr(), _lib.ptr(bias), y.data_ptr(), _lib.dtype_code(x), b, ih, iw, cx, ck, oh, ow,
            n, ldy, kh, kw, stride, pad, in_up, int(pixel_shuffle), wstride, _lib.stream_of(dev))
    _lib.check(code, "msg_conv2d_fprop")
    return y


def _launch_wgrad(gy, x, o, i, kh, kw, stride, pad, pixel_shuffle, per_sample, low_hw):
    dev = _lib.require_gpu(gy, x)
    gv, ldgy = _nhwc_view(gy)
    xv, cx = _nhwc_view(x)
    b, _, ih, iw = xv.shape
    oh, ow = low_hw if pixel_shuffle else gv.shape[2:]
    taps = kh * kw
    ldgw = _round_up(i, 4)
    if per_sample:
        gw = torch.empty((b, o, taps, ldgw), dtype=torch.float32, device=dev)
        k_chunks = 1
    else:
        gw = torch.zeros((o, taps, ldgw), dtype=torch.float32, device=dev)
        tiles = ((o + 127) // 128) * ((i + 127) // 128) * taps * b
        kp = 64 if x.dtype == torch.bfloat16 else 32
        k_chunks = max(1, min((oh * ow + 4 * kp - 1) // (4 * kp), (1024 + tiles - 1) // tiles))
        while b * k_chunks > 65535:
            k_chunks -= 1
    flops = 2.0 * b * oh * ow * o * i * taps
    with torch.cuda.device(dev), _lib.kernel_clock.span(f"conv_wgrad/{'bf16' if x.dtype == torch.bfloat16 else 'f32'}", flops):
        code = _lib.lib().msg_conv2d_wgrad(
            gv.data_ptr(), xv.data_ptr(), gw.data_ptr(), _lib.dtype_code(x), b, ih, iw, cx, i, oh, ow, ldgy, o, ldgw,
            kh, kw, stride, pad, int(pixel_shuffle), int(per_sample), k_chunks, _lib.stream_of(dev))
    _lib.check(code, "msg_conv2d_wgrad")
    gw = gw[..., :i]
    # [.., O, taps, I] -> [.., O, I, kh, kw]
    return gw.transpose(-1, -2).reshape(*gw.shape[:-2], i, kh, kw)


# ------------------------------------------------------------------------------------- the three primitives, raw
class Geometry:
    """kind 'conv': y = conv(x, w, stride, pad);  kind 'up2': y = conv_transpose(x, w^T, kernel 2, stride 2)."""
    __slots__ = ("kind", "kh", "kw", "stride", "pad", "x_hw", "y_hw", "per_sample")

    def __init__(self, kind, kh, kw, stride, pad, x_hw, per_sample):
        self.kind, self.kh, self.kw, self.stride, self.pad = kind, kh, kw, stride, pad
        self.x_hw, self.per_sample = tuple(x_hw), per_sample
        if kind == "up2":
            self.y_hw = (2 * x_hw[0], 2 * x_hw[1])
        else:
            self.y_hw = ((x_hw[0] + 2 * pad - kh) // stride + 1, (x_hw[1] + 2 * pad - kw) // stride + 1)


def _f_raw(x, w, bias, g: Geometry):
    wk, ck = _relay_fwd(w, x.dtype)
    o = w.shape[-4]
    if g.kind == "up2":
        # rows n = (2dy+dx)*O + o  <-  w[o, :, dy, dx]
        wk = wk.transpose(-3, -2).reshape(*wk.shape[:-3], 4 * o, 1, ck).contiguous()
        return _launch_fprop(x, wk, ck, None, 4 * o, g.x_hw, 1, 1, 1, 0, 1, True, g.per_sample, w.shape[-3])
    return _launch_fprop(x, wk, ck, bias, o, g.y_hw, g.kh, g.kw, g.stride, g.pad, 1, False, g.per_sample, w.shape[-3])


def _d_raw(gy, w, g: Geometry):
    i = w.shape[-3]
    if g.kind == "up2":
        wk, ok = _relay_dgrad(w, gy.dtype, flip=False)
        return _launch_fprop(gy, wk, ok, None, i, g.x_hw, 2, 2, 2, 0, 1, False, g.per_sample, w.shape[-4])
    wk, ok = _relay_dgrad(w, gy.dtype, flip=True)
    pad = g.kh - 1 - g.pad
    assert g.kh == g.kw
    return _launch_fprop(gy, wk, ok, None, i, g.x_hw, g.kh, g.kw, 1, pad, g.stride, False, g.per_sample, w.shape[-4])


def _g_raw(gy, x, o, i, g: Geometry):
    if g.kind == "up2":
        return _launch_wgrad(gy, x, o, i, 2, 2, 1, 0, True, g.per_sample, g.x_hw)
    return _launch_wgrad(gy, x, o, i, g.kh, g.kw, g.stride, g.pad, False, g.per_sample, None)


# ------------------------------------------------------------------------------------- autograd closure of F/D/G
class _ConvF(Function):
    @staticmethod
    def forward(ctx, x, w, bias, g):
        ctx.g = g
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w)
        return _f_raw(x, w, bias, g)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g = ctx.g
        gx = _ConvD.apply(gy, w, g) if ctx.needs_input_grad[0] else None
        gw = _ConvG.apply(gy, x, w.shape[-4], w.shape[-3], g) if ctx.needs_input_grad[1] else None
        gb = gy.float().sum(dim=(0, 2, 3)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


class _ConvD(Function):
    @staticmethod
    def forward(ctx, gy, w, g):
        ctx.g = g
        ctx.save_for_backward(gy, w)
        return _d_raw(gy, w, g)

    @staticmethod
    def backward(ctx, v):
        gy, w = ctx.saved_tensors
        g = ctx.g
        ggy = _ConvF.apply(v, w, None, g) if ctx.needs_input_grad[0] else None
        gw = _ConvG.apply(gy, v, w.shape[-4], w.shape[-3], g) if ctx.needs_input_grad[1] else None
        return ggy, gw, None


class _ConvG(Function):
    @staticmethod
    def forward(ctx, gy, x, o, i, g):
        ctx.g = g
        ctx.save_for_backward(gy, x)
        return _g_raw(gy, x, o, i, g)

    @staticmethod
    def backward(ctx, u):
        gy, x = ctx.saved_tensors
        g = ctx.g
        ggy = _ConvF.apply(x, u, None, g) if ctx.needs_input_grad[0] else None
        gx = _ConvD.apply(gy, u, g) if ctx.needs_input_grad[1] else None
        return ggy, gx, None, None, None


# ------------------------------------------------------------------------------------------------- public entry
def conv2d(x, weight, bias=None, stride=1, padding=0):
    """Shared-weight conv; weight [O,I,kh,kw] fp32 (already equalized-lr scaled), x [B,I,H,W]; y in x's dtype."""
    s = stride if isinstance(stride, int) else stride[0]
    p = padding if isinstance(padding, int) else padding[0]
    g = Geometry("conv", weight.shape[2], weight.shape[3], s, p, x.shape[2:], False)
    return _ConvF.apply(x, weight, None if bias is None else bias.float(), g)


def linear(x, weight, bias=None):
    """x [B,I] @ weight[O,I]^T (+ bias): the same contraction with the batch rows as 'pixels' of one sample."""
    b, i = x.shape
    o = weight.shape[0]
    y = conv2d(x.reshape(1, b, 1, i).permute(0, 3, 1, 2), weight.reshape(o, i, 1, 1), bias)
    return y.permute(0, 2, 3, 1).reshape(b, o)


def demod_coefficients(weight, style, scale):
    """d[b,o] = rsqrt(scale^2 * sum_i s[b,i]^2 * sum_k W[o,i,k]^2 + 1e-8)   (generator.py:384-388, refactored)."""
    wsq = weight[0].square().sum(dim=(2, 3))
    return torch.rsqrt((scale * scale) * (style.square() @ wsq.t()) + 1e-8)


def modulated_conv2d(x, weight, style, demodulate, upsample):
    """x [B,I,H,W]; weight [1,O,I,kh,kw] fp32; style [B,I] fp32 -> conv result (before any blur).

    One weight set per sample, w_b = d[b,o] * scale * W[o,i,k] * s[b,i] (multi_stylegan_generator.py:384-388), and
    one batched contraction (grid.z = sample) instead of the reference's groups=batch library conv.  The demodulation
    norm is evaluated from sum_k W^2 (an [O,I] table) so no [B,O,I,k,k] reduction pass is needed.
    """
    _lib.require_gpu(x, weight, style)
    _, out_c, in_c, kh, kw = weight.shape
    scale = math.sqrt(2.0) / math.sqrt(in_c * kh * kw)
    wmod = (weight * scale) * style[:, None, :, None, None]
    if demodulate:
        wmod = wmod * demod_coefficients(weight, style, scale)[:, :, None, None, None]
    g = Geometry("up2" if upsample else "conv", kh, kw, 1, kh // 2, x.shape[2:], True)
    return _ConvF.apply(x, wmod, None, g)
