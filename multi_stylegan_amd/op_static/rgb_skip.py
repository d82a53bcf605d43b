"""The generator's RGB skip path as one launch per level and direction (csrc/rgb_skip.hip):

    out = float(conv) + bias + upfirdn2d(skip, fir, up=2, pad=(2, 1))

-- OutputBlock.forward of the reference (multi_stylegan_generator.py:513-523 with Upsample :545-574).  `conv` is the thin
1x1 modulated conv's channels-last result (bf16 or fp32 storage), `bias` a per-channel fp32 vector (the heads' scalar biases
expanded), `skip` the previous level's fp32 planes.  The op is linear in all three, so its backward is one gather launch and
its double backward (path-length regulariser) is the forward applied to the cotangents: both Functions below are
differentiable to any order.
"""
from typing import Optional

import torch
from torch.autograd import Function

from .. import _lib
from .._autograd import _derive


def supported(conv: torch.Tensor, skip: Optional[torch.Tensor], fir: Optional[torch.Tensor], up: int, pad) -> bool:
    if not (conv.is_cuda and conv.ndim == 4 and conv.dtype in (torch.float32, torch.bfloat16)):
        return False
    b, c, h, w = conv.shape
    if c > 8 or w % 4 or h % 2:
        return False
    if skip is None:
        return True
    return (fir is not None and tuple(fir.shape) == (4, 4) and fir.dtype == torch.float32 and up == 2
            and tuple(pad) == (2, 1) and skip.dtype == torch.float32 and tuple(skip.shape) == (b, c, h // 2, w // 2))


def _launch_fwd(conv, bias, skip, fir):
    from .. import conv_ops
    dev = _lib.require_gpu(conv, bias, skip, fir)
    b, c, h, w = conv.shape
    cv, ld = conv_ops._nhwc_view(conv)
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
    skip_c = skip.contiguous() if skip is not None else None
    bias_c = bias.contiguous() if bias is not None else None
    nbytes = conv.numel() * conv.element_size() + out.numel() * 4 + (skip.numel() * 4 if skip is not None else 0)
    with _lib.on_device(dev), _lib.kernel_clock.span("rgb_skip_merge/f32", nbytes):
        code = _lib.lib().msg_rgb_skip_merge(cv.data_ptr(), _lib.dtype_code(conv), ld, _lib.ptr(bias_c), _lib.ptr(skip_c),
                                             _lib.ptr(fir if skip is not None else None), out.data_ptr(), b, c, h, w,
                                             _lib.stream_of(dev))
    _lib.check(code, "msg_rgb_skip_merge")
    return out


def _launch_bwd(g, conv_dtype, want_conv, want_skip, fir):
    from .. import conv_ops
    dev = _lib.require_gpu(g, fir)
    b, c, h, w = g.shape
    g = g.contiguous()
    g_conv, ld = (conv_ops._alloc_out(b, c, h, w, conv_dtype, dev) if want_conv else (None, 8))
    g_skip = torch.empty((b, c, h // 2, w // 2), dtype=torch.float32, device=dev) if want_skip else None
    nbytes = g.numel() * 4 + (g_conv.numel() * g_conv.element_size() if want_conv else 0) + \
        (g_skip.numel() * 4 if want_skip else 0)
    with _lib.on_device(dev), _lib.kernel_clock.span("rgb_skip_merge_bwd/f32", nbytes):
        code = _lib.lib().msg_rgb_skip_merge_backward(g.data_ptr(), _lib.ptr(g_conv), _lib.dtype_code(
            g_conv if want_conv else g), ld, _lib.ptr(g_skip), _lib.ptr(fir if want_skip else None), b, c, h, w,
            _lib.stream_of(dev))
    _lib.check(code, "msg_rgb_skip_merge_backward")
    return g_conv, g_skip


class _RgbSkipMerge(Function):
    @staticmethod
    def forward(ctx, conv, bias, skip, fir):
        ctx.save_for_backward(fir)
        ctx.meta = (conv.dtype, skip is not None, bias is not None)
        return _launch_fwd(conv, bias, skip, fir)

    @staticmethod
    def backward(ctx, g):
        fir, = ctx.saved_tensors
        conv_dtype, has_skip, has_bias = ctx.meta
        need = ctx.needs_input_grad
        g_conv, g_bias, g_skip = _derive(_RgbSkipMergeBackward, g, fir, conv_dtype, need[0], has_bias and need[1],
                                                             has_skip and need[2])
        return g_conv if need[0] else None, g_bias if has_bias and need[1] else None, \
            g_skip if has_skip and need[2] else None, None


class _RgbSkipMergeBackward(Function):
    """g -> (g_conv, g_bias, g_skip); its own backward is the forward merge of the three cotangents."""

    @staticmethod
    def forward(ctx, g, fir, conv_dtype, want_conv, want_bias, want_skip):
        ctx.save_for_backward(fir)
        ctx.meta = (want_conv, want_bias, want_skip, g.shape)
        g_conv, g_skip = (None, None)
        if want_conv or want_skip:
            g_conv, g_skip = _launch_bwd(g, conv_dtype, want_conv, want_skip, fir)
        g_bias = g.sum(dim=(0, 2, 3)) if want_bias else None
        zero = g.new_zeros(0)
        outs = (g_conv if want_conv else zero, g_bias if want_bias else zero, g_skip if want_skip else zero)
        ctx.mark_non_differentiable(*[o for o, wanted in zip(outs, (want_conv, want_bias, want_skip)) if not wanted])
        return outs

    @staticmethod
    def backward(ctx, gg_conv, gg_bias, gg_skip):
        fir, = ctx.saved_tensors
        want_conv, want_bias, want_skip, shape = ctx.meta
        b, c, h, w = shape
        if not want_conv:
            # (no conv cotangent: the merge kernel still needs its first operand)
            gg_conv = torch.zeros((b, c, h, w), dtype=torch.float32, device=fir.device).contiguous(
                memory_format=torch.channels_last)
        gg = _RgbSkipMerge.apply(gg_conv, gg_bias if want_bias else None, gg_skip if want_skip else None, fir)
        return gg, None, None, None, None, None


def rgb_skip_merge(conv: torch.Tensor, bias: Optional[torch.Tensor], skip: Optional[torch.Tensor],
                   fir: Optional[torch.Tensor]) -> torch.Tensor:
    """conv [B, C, H, W] (channels-last storage, C <= 8), bias fp32 [C] or None, skip fp32 [B, C, H/2, W/2] or None, fir the
    [4, 4] FIR of the Upsample module -> fp32 planes [B, C, H, W]."""
    return _RgbSkipMerge.apply(conv, bias, skip, fir)
