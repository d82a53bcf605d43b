"""Dense contractions of the hot path: shared-weight conv, modulated/demodulated conv, linear.

Stage 1 of the build routes the contraction itself through the ROCm libraries that ship inside torch
(MIOpen / rocBLAS on the GPU) while the hand-written MFMA kernels of csrc/ are brought up; everything else of
the path (FIR resampling, bias/noise/activation) already runs on the repo's own HIP kernels.  Inputs must live
on the GPU: nothing here falls back to the CPU.
"""
import math

import torch
import torch.nn.functional as F

from . import _lib


def to_compute_layout(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """Feature maps with >= 8 channels are kept channels-last (NHWC in HBM, NCHW logical shape)."""
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    if x.ndim == 4 and x.shape[1] >= 8:
        return x.contiguous(memory_format=torch.channels_last)
    return x.contiguous()


def conv2d(x, weight, bias=None, stride=1, padding=0):
    """Plain conv with an already equalized-lr-scaled fp32 weight [O,I,kh,kw]; output in x's dtype/layout."""
    _lib.require_gpu(x, weight)
    b = None if bias is None else bias.to(x.dtype)
    return to_compute_layout(F.conv2d(x, weight.to(x.dtype), b, stride=stride, padding=padding))


def linear(x, weight, bias=None):
    _lib.require_gpu(x, weight)
    return F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


def demod_coefficients(weight, style, scale):
    """d[b,o] = rsqrt(scale^2 * sum_i s[b,i]^2 * sum_k W[o,i,k]^2 + 1e-8)   (generator.py:384-388, refactored)."""
    wsq = weight[0].square().sum(dim=(2, 3))                                  # [O, I]
    return torch.rsqrt((scale * scale) * (style.square() @ wsq.t()) + 1e-8)     # [B, O]


def modulated_conv2d(x, weight, style, demodulate, upsample):
    """x [B,I,H,W]; weight [1,O,I,kh,kw] fp32; style [B,I] fp32.  Returns the conv result (before any blur).

    Same function as multi_stylegan_generator.py:384-411 evaluated as
        y[b,o] = d[b,o] * sum_{i,k} (scale*W[o,i,k]) * (s[b,i] * x[b,i, . + k])
    so that one weight tensor serves the whole batch; the 2x2 stride-2 transposed conv of the up-sampling layers
    has no overlap between taps (kernel = stride), i.e. four independent 1x1 products written pixel-shuffled.
    """
    _lib.require_gpu(x, weight, style)
    _, out_c, in_c, kh, kw = weight.shape
    scale = math.sqrt(2.0) / math.sqrt(in_c * kh * kw)
    w = weight[0] * scale
    xs = x * style.to(x.dtype)[:, :, None, None]
    if upsample:
        y = F.conv_transpose2d(xs, w.transpose(0, 1).to(x.dtype), stride=2, padding=0)
    else:
        y = F.conv2d(xs, w.to(x.dtype), padding=(kh // 2, kw // 2))
    if demodulate:
        d = demod_coefficients(weight, style, scale)
        y = y * d.to(x.dtype)[:, :, None, None]
    return to_compute_layout(y)
