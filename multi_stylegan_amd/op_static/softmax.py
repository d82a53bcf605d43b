"""Row softmax of the non-local block's attention map on the gfx950 kernel (csrc/softmax.hip), twice differentiable.

Reference call site: multi_stylegan/u_net_2d_discriminator.py:378 (``F.softmax(torch.bmm(theta^T, phi), -1)``).
One wave keeps a row in registers: forward reads and writes the map once, backward reads two maps and writes one.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .._autograd import _derive

_MAX_COLS = 4096


def _supported(x):
    vec = 16 // x.element_size()
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % vec == 0 and \
        x.shape[-1] <= _MAX_COLS and x.numel() > 0


def _call(name, nbytes, *tensors):
    first = tensors[0]
    dev = _lib.require_gpu(*tensors)
    cols = first.shape[-1]
    rows = first.numel() // cols
    with _lib.on_device(dev), _lib.kernel_clock.span((name, first.dtype), nbytes):
        code = getattr(_lib.lib(), f"msg_{name}")(*[t.data_ptr() for t in tensors], _lib.dtype_code(first), rows, cols,
                                                  _lib.stream_of(dev))
    _lib.check(code, f"msg_{name}")


class _SoftmaxRowsBackward(Function):
    @staticmethod
    def forward(ctx, y, gy):
        y, gy = y.contiguous(), gy.contiguous()
        gx = torch.empty_like(y)
        _call("softmax_rows_backward", 3 * y.numel() * y.element_size(), y, gy, gx)
        ctx.save_for_backward(y, gy)
        return gx

    @staticmethod
    def backward(ctx, v):
        # gx = y * (gy - <gy, y>): second-order terms (R1 through the discriminator)
        y, gy = ctx.saved_tensors
        if not torch.is_grad_enabled():
            # closed form, row by row with s = <gy, y>, t = <v, y>:
            #   d gx / d gy  applied to v:  y * (v - t)          -- the first-order kernel itself, with v in gy's place
            #   d gx / d y   applied to v:  v * (gy - s) - gy * t
            # (the composite below needs ~15 fp32 passes over the [B, HW, HW/4] map for the same two results; round 5: both in
            #  ONE launch, msg_softmax_rows_backward2 -- three maps read, two written -- where rounds 3-4 still ran two row
            #  sums, an addcmul and five elementwise passes beside the first-order kernel)
            v = v.contiguous()
            d_y, d_gy = torch.empty_like(y), torch.empty_like(y)
            _call("softmax_rows_backward2", 5 * y.numel() * y.element_size(), y, gy, v, d_y, d_gy)
            return d_y, d_gy
        with torch.enable_grad():                       # third and higher order: differentiate the composite formulation
            y_, gy_ = y.detach().requires_grad_(True), gy.detach().requires_grad_(True)
            yf, gf = y_.float(), gy_.float()
            gx = (yf * (gf - (gf * yf).sum(dim=-1, keepdim=True))).to(y.dtype)
            d_y, d_gy = torch.autograd.grad(gx, (y_, gy_), v, create_graph=True)
        return d_y, d_gy


class _SoftmaxRows(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        _call("softmax_rows", 2 * x.numel() * x.element_size(), x, y)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        return _derive(_SoftmaxRowsBackward, y, gy)


def softmax_rows(x: torch.Tensor) -> torch.Tensor:
    """softmax over the last dimension (fp32 arithmetic, result in x's dtype).  Rows longer than 4096 columns or not a
    multiple of the 16-byte vector go to the ROCm library softmax (same device, same semantics)."""
    if not _supported(x):
        return torch.softmax(x, dim=-1)
    return _SoftmaxRows.apply(x)
