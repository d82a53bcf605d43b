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

    def forward_activated(self, input: torch.Tensor, activation, grad_slot=None) -> torch.Tensor:
        """``activation(self(input))`` for a following FusedLeakyReLU, its bias + leaky ReLU fused into this conv's
        epilogue (one pass over the output map instead of two; same result bit for bit)."""
        if self.bias is not None or not input.is_cuda or not conv_ops.FUSE_ACTIVATION:
            return activation(self(input))
        return conv_ops.conv2d_bias_act(input, self.weight, activation.bias, stride=self.stride, padding=self.padding,
                                        wscale=self.scale, negative_slope=activation.negative_slope,
                                        scale=activation.scale, grad_slot=grad_slot)


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
