"""The discriminator's pixel-wise head, FusedLeakyReLU(C) -> bias-free 1x1 EqualizedConv2d(C, 1), as one streaming pass per
direction (csrc/pointwise_head.hip; reference u_net_2d_discriminator.py:93-97).  First-order backward on the kernel; a
backward that is itself differentiated (the R1 regulariser's create_graph pass) re-derives itself from the two-op form,
which is differentiable to any order.
"""
import os

import torch
from torch.autograd import Function

from .. import _lib

POINTWISE_HEAD = bool(int(os.environ.get("MSG_POINTWISE_HEAD", "1")))        # 0 / False: the two-op form (A/B; tests compare)


def supported(x: torch.Tensor, conv, act) -> bool:
    if not (POINTWISE_HEAD and x.is_cuda and x.ndim == 4 and x.dtype == torch.bfloat16):
        return False
    c = x.shape[1]
    lanes = c // 8
    if c % 8 or c > 512 or lanes & (lanes - 1) or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    w = conv.weight
    return conv.bias is None and tuple(w.shape) == (1, c, 1, 1) and tuple(conv.stride) == (1, 1) and \
        tuple(conv.padding) == (0, 0) and act.bias is not None and act.bias.shape == (c,)


def _two_op(x, act_bias, weight, wscale, alpha, scale):
    from .. import conv_ops
    from .fused_act import fused_leaky_relu
    return conv_ops.conv2d(fused_leaky_relu(x, act_bias, alpha, scale), weight, None, wscale=wscale).float()


class _ActPointwiseHead(Function):
    @staticmethod
    def forward(ctx, x, act_bias, weight, wscale, alpha, scale):
        dev = _lib.require_gpu(x, act_bias, weight)
        b, c, h, w = x.shape
        y = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        b32 = act_bias.detach().float().contiguous()
        w32 = weight.detach().float().reshape(c).contiguous()
        with _lib.on_device(dev), _lib.kernel_clock.span(("pointwise_head_fwd", x.dtype), x.numel() * 2 + y.numel() * 4):
            code = _lib.lib().msg_act_pointwise_head(x.data_ptr(), b32.data_ptr(), w32.data_ptr(), y.data_ptr(), _lib.MSG_BF16,
                                                     b * h * w, c, float(wscale), float(alpha), float(scale), _lib.stream_of(dev))
        _lib.check(code, "msg_act_pointwise_head")
        ctx.save_for_backward(x, act_bias, weight)
        ctx.cfg = (float(wscale), float(alpha), float(scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, act_bias, weight = ctx.saved_tensors
        wscale, alpha, scale = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # a differentiated backward (R1): through the two-op form
            with torch.enable_grad():
                ins = [t for t, n in zip((x, act_bias, weight), need[:3]) if n]
                grads = list(torch.autograd.grad(_two_op(x, act_bias, weight, wscale, alpha, scale), ins, gy, create_graph=True,
                                                 allow_unused=True))
            out = [grads.pop(0) if n else None for n in need[:3]]
            return out[0], out[1], out[2], None, None, None
        dev = x.device
        b, c, h, w = x.shape
        g32 = gy.detach().float().contiguous()
        gx = torch.empty_like(x)
        sums = torch.empty(2 * c, dtype=torch.float32, device=dev)
        need_ws = _lib.lib().msg_act_pointwise_head_backward_workspace(b * h * w, c)
        ws = torch.empty(need_ws, dtype=torch.float32, device=dev)
        b32 = act_bias.detach().float().contiguous()
        w32 = weight.detach().float().reshape(c).contiguous()
        with _lib.on_device(dev), _lib.kernel_clock.span(("pointwise_head_bwd", x.dtype), x.numel() * 4 + g32.numel() * 4):
            code = _lib.lib().msg_act_pointwise_head_backward(
                x.data_ptr(), b32.data_ptr(), w32.data_ptr(), g32.data_ptr(), gx.data_ptr(), sums.data_ptr(), _lib.MSG_BF16,
                b * h * w, c, wscale, alpha, scale, ws.data_ptr(), need_ws, _lib.stream_of(dev))
        _lib.check(code, "msg_act_pointwise_head_backward")
        return (gx if need[0] else None), (sums[:c] if need[1] else None), \
            (sums[c:].reshape(weight.shape) if need[2] else None), None, None, None


def act_pointwise_head(x, act, conv):
    """conv(act(x)).float() for act = FusedLeakyReLU(C), conv = EqualizedConv2d(C, 1, 1x1, bias=False) -> [B, 1, H, W] fp32."""
    return _ActPointwiseHead.apply(x, act.bias, conv.weight, conv.scale, act.negative_slope, act.scale)
