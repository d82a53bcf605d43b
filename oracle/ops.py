"""Oracle restatement of the hot-path operators (plain differentiable torch ops).

Because everything here is composed of stock torch ops, first- and
second-order gradients come from torch autograd itself and are the yardstick
for the hand-written backward / double-backward kernels.
"""
import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# a1: upfirdn2d  (multi_stylegan/op_static/upfirdn2d_kernel.cu:52-137,
#                 multi_stylegan/op_static/upfirdn2d.py:91-153)
# --------------------------------------------------------------------------
def upfirdn_out_size(n: int, up: int, down: int, pad0: int, pad1: int, taps: int) -> int:
    """Output extent; upfirdn2d_kernel.cu:167-168, upfirdn2d.py:105-106."""
    return (n * up + pad0 + pad1 - taps) // down + 1


def upfirdn2d(x: torch.Tensor, fir: torch.Tensor, up: int = 1, down: int = 1,
              pad: Sequence[int] = (0, 0)) -> torch.Tensor:
    """Zero-insert by ``up``, pad/crop, TRUE convolution with ``fir``, keep every ``down``-th.

    x is [B, C, H, W]; pad = (before, after) applied to both axes as the
    reference's public wrapper does (upfirdn2d.py:148-153).  The kernel sums
    input * flipped-FIR (upfirdn2d_kernel.cu:77,128-131), i.e. a true
    convolution, which ``F.conv2d`` (a correlation) performs with the FIR
    flipped once more.
    """
    b, c, h, w = x.shape
    p0, p1 = int(pad[0]), int(pad[1])
    planes = x.reshape(b * c, 1, h, w)
    if up > 1:
        grid = planes.new_zeros(b * c, 1, h * up, w * up)
        grid[:, :, ::up, ::up] = planes          # sample sits first in each up-cell
        planes = grid
    planes = F.pad(planes, [p0, p1, p0, p1])     # negative values crop
    taps = torch.flip(fir, [0, 1]).to(planes.dtype)[None, None]
    full = F.conv2d(planes, taps)
    out = full[:, :, ::down, ::down]
    return out.reshape(b, c, out.shape[-2], out.shape[-1])


def upfirdn2d_scalar(x, fir, up, down, pad0, pad1):
    """Literal per-output restatement of the CUDA index math (tiny inputs only).

    upfirdn2d_kernel.cu:114-133: mid = o*down + up-1-pad0; in = floor(mid/up);
    k = (in+1)*up - mid - 1; v = sum sx[in+y][in+x] * flipped_fir[k_y + y*up][k_x + x*up].
    x: [P, H, W] nested lists / tensors; returns [P, OH, OW] tensor (float64).
    """
    x = torch.as_tensor(x, dtype=torch.float64)
    fir = torch.as_tensor(fir, dtype=torch.float64)
    kh, kw = fir.shape
    p, h, w = x.shape
    oh = upfirdn_out_size(h, up, down, pad0, pad1, kh)
    ow = upfirdn_out_size(w, up, down, pad0, pad1, kw)
    flipped = torch.flip(fir, [0, 1])
    out = torch.zeros(p, oh, ow, dtype=torch.float64)
    for oy in range(oh):
        mid_y = oy * down + up - 1 - pad0
        in_y = mid_y // up                      # python // is floor division
        ky = (in_y + 1) * up - mid_y - 1
        for ox in range(ow):
            mid_x = ox * down + up - 1 - pad0
            in_x = mid_x // up
            kx = (in_x + 1) * up - mid_x - 1
            acc = torch.zeros(p, dtype=torch.float64)
            for yy in range(kh // up):
                for xx in range(kw // up):
                    sy, sx = in_y + yy, in_x + xx
                    if 0 <= sy < h and 0 <= sx < w:
                        acc += x[:, sy, sx] * flipped[ky + yy * up, kx + xx * up]
            out[:, oy, ox] = acc
    return out


def make_fir(taps: Sequence[float] = (1, 3, 3, 1), gain: float = 1.0) -> torch.Tensor:
    """Separable FIR normalised to sum ``gain``; multi_stylegan_generator.py:553-566,619-632."""
    t = torch.tensor(list(taps), dtype=torch.float32)
    k = t[None, :] * t[:, None]
    k = k / k.sum()
    return k * gain if gain != 1.0 else k


# --------------------------------------------------------------------------
# a2: fused bias + leaky-ReLU (+ noise)
#     (op_static/fused_bias_act_kernel.cu:18-49, op_static/fused_act.py:22-89,
#      multi_stylegan_generator.py:267-292)
# --------------------------------------------------------------------------
def fused_leaky_relu(x: torch.Tensor, bias: Optional[torch.Tensor], negative_slope: float = 0.2,
                     scale: float = 2 ** 0.5) -> torch.Tensor:
    """y = lrelu(x + b[c]) * scale; the gradient mask is sign(out) (fused_act.py:31-33).

    torch's leaky_relu uses (x > 0 ? 1 : slope) which coincides with the
    reference's (out > 0 ? 1 : slope) for scale > 0, including at exactly 0.
    """
    if bias is not None and bias.numel():
        x = x + bias.view(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x, negative_slope) * scale


def noise_injection(x: torch.Tensor, weight: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """x + w * noise[b,0,h,w]; multi_stylegan_generator.py:288-292."""
    return x + weight * noise


# --------------------------------------------------------------------------
# a4: equalized-lr layers and pixel norm (equalized_layer.py:9-74,210-277)
# --------------------------------------------------------------------------
def eq_scale(fan_in: int) -> float:
    return math.sqrt(2.0) / math.sqrt(fan_in)


def equalized_linear(x, weight, bias=None):
    """equalized_layer.py:244-254: W*sqrt(2/in), b*sqrt(2/out)."""
    out_f, in_f = weight.shape
    b = None if bias is None else bias * eq_scale(out_f)
    return F.linear(x, weight * eq_scale(in_f), b)


def equalized_conv2d(x, weight, bias=None, stride=1, padding=0):
    """equalized_layer.py:63-74: W*sqrt(2/(in*kh*kw)), b*sqrt(2/out)."""
    out_c, in_c, kh, kw = weight.shape
    # the reference keeps the scales as float32 tensors (equalized_layer.py:42-44)
    s_w = torch.tensor(math.sqrt(2.0) / math.sqrt(in_c * kh * kw)).float().to(weight.dtype)
    b = None
    if bias is not None:
        b = bias * torch.tensor(math.sqrt(2.0) / math.sqrt(out_c)).float().to(weight.dtype)
    return F.conv2d(x, weight * s_w, b, stride=stride, padding=padding)


def equalized_conv_transpose2d(x, weight, bias=None, stride=2, padding=0):
    """equalized_layer.py:127-143: F.conv_transpose2d with W * sqrt(2 / (in * kh * kw)), b * sqrt(2 / out); weight
    [in, out, kh, kw]."""
    scale = math.sqrt(2.0) / math.sqrt(weight.shape[0] * weight.shape[2] * weight.shape[3])
    b = None if bias is None else bias * (math.sqrt(2.0) / math.sqrt(weight.shape[1]))
    return F.conv_transpose2d(x, weight * scale, b, stride=stride, padding=padding)


def equalized_conv1d(x, weight, bias=None, stride=1, padding=1):
    """equalized_layer.py:192-207: F.conv1d with W * sqrt(2 / (in * k)), b * sqrt(2 / out); weight [out, in, k]."""
    scale = math.sqrt(2.0) / math.sqrt(weight.shape[1] * weight.shape[2])
    b = None if bias is None else bias * (math.sqrt(2.0) / math.sqrt(weight.shape[0]))
    return F.conv1d(x, weight * scale, b, stride=stride, padding=padding)


def pixel_norm(x, alpha: float = 1e-8):
    """equalized_layer.py:276."""
    return x / torch.sqrt(torch.mean(x * x, dim=1, keepdim=True) + alpha)


# --------------------------------------------------------------------------
# a3: dual-styled modulated / demodulated conv
#     (multi_stylegan_generator.py:365-414)
# --------------------------------------------------------------------------
def modulated_conv2d(x: torch.Tensor, weight: torch.Tensor, style: torch.Tensor, *,
                     demodulate: bool, upsample: bool,
                     blur_fir: Optional[torch.Tensor] = None,
                     blur_pad: Tuple[int, int] = (2, 1)) -> torch.Tensor:
    """x [B,I,H,W], weight [1,O,I,kh,kw], style [B,I] (already the modulated style s).

    w' = sqrt(2/(I*kh*kw)) * W * s  (:384); demod by rsqrt(sum_{i,k} w'^2 + 1e-8)
    (:386-388); then one conv per sample -- 3x3/1x1 stride 1 "same" (:406-411)
    or 2x2 stride-2 transposed conv followed by the 4x4 FIR blur (:391-403).
    """
    bsz, in_c, h, w = x.shape
    _, out_c, _, kh, kw = weight.shape
    scale = math.sqrt(2.0) / math.sqrt(in_c * kh * kw)
    wmod = (scale * weight) * style.reshape(bsz, 1, in_c, 1, 1)
    if demodulate:
        wmod = wmod * torch.rsqrt(wmod.square().sum(dim=(2, 3, 4), keepdim=True) + 1e-8)
    flat = x.reshape(1, bsz * in_c, h, w)
    if upsample:
        wt = wmod.transpose(1, 2).reshape(bsz * in_c, out_c, kh, kw)
        y = F.conv_transpose2d(flat, wt, stride=2, padding=0, groups=bsz)
        y = y.reshape(bsz, out_c, y.shape[-2], y.shape[-1])
        return upfirdn2d(y, blur_fir, pad=blur_pad)
    y = F.conv2d(flat, wmod.reshape(bsz * out_c, in_c, kh, kw), padding=(kh // 2, kw // 2), groups=bsz)
    return y.reshape(bsz, out_c, y.shape[-2], y.shape[-1])


# --------------------------------------------------------------------------
# a5/a6: discriminator helpers (u_net_2d_discriminator.py:205-217,359-381)
# --------------------------------------------------------------------------
def minibatch_stddev(x, alpha: float = 1e-8):
    """One scalar plane: mean over (c,h,w) of the per-position batch std (:211-216)."""
    dev = x - x.mean(dim=0, keepdim=True)
    std = torch.sqrt(dev.square().mean(dim=0).clamp(min=alpha))
    plane = std.mean().reshape(1, 1, 1, 1).expand(x.shape[0], 1, x.shape[2], x.shape[3])
    return torch.cat([x, plane], dim=1)


def non_local_attention(theta, phi, g):
    """softmax(theta^T phi) applied to g; theta [B,c8,HW], phi [B,c8,HW/4], g [B,c2,HW/4] (:378-380)."""
    beta = torch.softmax(torch.bmm(theta.transpose(1, 2), phi), dim=-1)
    return torch.bmm(g, beta.transpose(1, 2))
