"""Equalized-learning-rate layers with the reference's class names, constructor arguments and parameter
names (multi_stylegan/equalized_layer.py:9-277).  The runtime weight/bias scaling (W*sqrt(2/fan_in),
b*sqrt(2/out)) is folded into the contraction call instead of being materialised as a module op."""
import math
from typing import Tuple, Union

import torch
import torch.nn as nn

from . import conv_ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class EqualizedConv2d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[int, Tuple[int, int]] = 3,
                 stride: Union[int, Tuple[int, int]] = 1, padding: Union[int, Tuple[int, int]] = 1,
                 bias: bool = True):
        super().__init__()
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.scale = math.sqrt(2.0) / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
        self.scale_bias = math.sqrt(2.0) / math.sqrt(out_channels)

    def extra_repr(self):
        o, i, kh, kw = self.weight.shape
        return f"{i}, {o}, kernel_size=({kh}, {kw}), stride={self.stride}, padding={self.padding}, " \
               f"bias={self.bias is not None}"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        b = None if self.bias is None else self.bias * self.scale_bias
        return conv_ops.conv2d(input, self.weight, b, stride=self.stride, padding=self.padding, wscale=self.scale)

    def forward_activated(self, input: torch.Tensor, activation, grad_slot=None, out_grad_scale=None, act_handle=None,
                          input_act=None) -> torch.Tensor:
        """``activation(self(input))`` for a following FusedLeakyReLU, its bias + leaky ReLU fused into this conv's
        epilogue (one pass over the output map instead of two; same result bit for bit)."""
        if self.bias is not None or not input.is_cuda:
            return activation(self(input))
        return conv_ops.conv2d_bias_act(input, self.weight, activation.bias, stride=self.stride, padding=self.padding,
                                        wscale=self.scale, negative_slope=activation.negative_slope,
                                        scale=activation.scale, grad_slot=grad_slot, out_grad_scale=out_grad_scale,
                                        act_handle=act_handle, input_act=input_act)


class EqualizedTransposedConv2d(nn.Module):
    """Reference equalized_layer.py:77-143 (weight [in, out, kh, kw], bias initialised to ones).  Not instantiated by
    the generator or the discriminator; provided for the public surface in the form the reference's defaults use --
    kernel 2, stride 2, padding 0, the non-overlapping transposed conv -- on the sub-pixel contraction kernels.  Other
    geometries raise."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[int, Tuple[int, int]] = 2,
                 stride: Union[int, Tuple[int, int]] = 2, padding: Union[int, Tuple[int, int]] = 0,
                 bias: bool = True) -> None:
        super().__init__()
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.weight = nn.Parameter(torch.randn(in_channels, out_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.ones(out_channels)) if bias else None
        self.scale = math.sqrt(2.0) / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
        self.scale_bias = math.sqrt(2.0) / math.sqrt(out_channels)

    def extra_repr(self):
        i, o, kh, kw = self.weight.shape
        return f"{o}, {i}, kernel_size=({kh}, {kw}), stride={self.stride}, padding={self.padding}, " \
               f"bias={self.bias is not None}"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0):
            from ._lib import MsgHipError
            raise MsgHipError("EqualizedTransposedConv2d: only kernel_size 2, stride 2, padding 0 is implemented")
        b = None if self.bias is None else self.bias * self.scale_bias
        return conv_ops.conv_transpose2d_2x2(conv_ops.to_compute_layout(input), self.weight, b, wscale=self.scale)


class EqualizedConv1d(nn.Module):
    """Reference equalized_layer.py:146-207 (weight [out, in, k], bias initialised to ones).  Not instantiated by the
    models; runs as a 1x1 contraction over the tap-gathered signal (k shifted, strided views stacked along the
    channels), the form the few-channel 2-D convs use."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, padding: int = 1,
                 bias: bool = True) -> None:
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.ones(out_channels)) if bias else None
        self.scale = math.sqrt(2.0) / math.sqrt(in_channels * kernel_size)
        self.scale_bias = math.sqrt(2.0) / math.sqrt(out_channels)

    def extra_repr(self):
        o, i, k = self.weight.shape
        return f"{i}, {o}, kernel_size={k}, stride={self.stride}, padding={self.padding}, bias={self.bias is not None}"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        bsz, channels, length = input.shape
        k, s, p = self.kernel_size, self.stride, self.padding
        out_len = (length + 2 * p - k) // s + 1
        padded = torch.nn.functional.pad(input, (p, p))
        taps = torch.cat([padded[:, :, t:t + s * (out_len - 1) + 1:s] for t in range(k)], dim=1)    # [B, k*C, L_out]
        weight = self.weight.permute(0, 2, 1).reshape(self.weight.shape[0], k * channels, 1, 1)      # tap-major, as `taps`
        b = None if self.bias is None else self.bias * self.scale_bias
        y = conv_ops.conv2d(conv_ops.to_compute_layout(taps.unsqueeze(2)), weight, b, stride=1, padding=0,
                            wscale=self.scale)
        return y.squeeze(2)


class EqualizedLinear(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bias: bool = True) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.scale = math.sqrt(2.0) / math.sqrt(in_channels)
        self.scale_bias = math.sqrt(2.0) / math.sqrt(out_channels)

    def extra_repr(self):
        return f"{self.weight.shape[1]}, {self.weight.shape[0]}, bias={self.bias is not None}"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if input.is_cuda:                       # both equalized-lr gains ride inside the kernel (no bias * scale launch)
            return conv_ops.linear(input, self.weight, self.bias, wscale=self.scale, bias_scale=self.scale_bias)
        b = None if self.bias is None else self.bias * self.scale_bias
        return conv_ops.linear(input, self.weight, b, wscale=self.scale)


class PixelwiseNormalization(nn.Module):
    def __init__(self, alpha: float = 1e-8) -> None:
        super().__init__()
        self.alpha = alpha

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return input * torch.rsqrt(torch.mean(input * input, dim=1, keepdim=True) + self.alpha)
