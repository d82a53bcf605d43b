"""2x2 / stride-2 max-pooling of channels-last maps on the gfx950 kernels (csrc/maxpool.hip) -- the pooling of the NonLocalBlock's
key and value maps (reference u_net_2d_discriminator.py:366-370: F.max_pool2d(phi(x)), F.max_pool2d(g(x)))."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from .._autograd import _derive

def _pitch(x):
    """Pixel pitch of a channels-last [B,C,H,W] map (or channel-slice of one) the kernel takes as it is, else None."""
    b, c, h, w = x.shape
    sb, sc, sh, sw = x.stride()
    vec = 16 // x.element_size()
    ok = (sc == 1 or c == 1) and sw >= c and sw % vec == 0 and sh == w * sw and (sb == h * w * sw or b == 1) and \
        x.data_ptr() % 16 == 0
    return sw if ok else None


class _MaxPool2x2(Function):
    @staticmethod
    def forward(ctx, x):
        dev = _lib.require_gpu(x)
        b, c, h, w = x.shape
        y = torch.empty((b, c, h // 2, w // 2), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        vec = 16 // x.element_size()
        idx = torch.empty(y.numel() // vec, dtype=torch.int16, device=dev) if ctx.needs_input_grad[0] else None
        with _lib.on_device(dev):
            code = _lib.lib().msg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), _lib.ptr(idx), _lib.dtype_code(x), b, h, w, c,
                                                 _pitch(x), _lib.stream_of(dev))
        _lib.check(code, "msg_maxpool2x2_fwd")
        ctx.save_for_backward(x, idx)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, idx = ctx.saved_tensors
        return _derive(_MaxPool2x2Backward, gy, idx, x.shape)


class _MaxPool2x2Backward(Function):
    """gx = gy routed to the window positions the forward chose (2 bits per element), zero elsewhere.  Linear in gy with a
    routing that does not depend on it, so its own derivative -- R1 differentiates the discriminator's backward -- is the
    GATHER of the cotangent at the same positions (msg_maxpool2x2_gather), to any order; rounds 3-4 re-ran the library's
    pooling on the saved input to get a differentiable graph."""

    @staticmethod
    def forward(ctx, gy, idx, x_shape):
        b, c, h, w = x_shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((b, c, h, w), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        with _lib.on_device(gy.device):
            code = _lib.lib().msg_maxpool2x2_bwd(gy.data_ptr(), idx.data_ptr(), gx.data_ptr(), _lib.dtype_code(gy), b, h, w, c,
                                                 _lib.stream_of(gy.device))
        _lib.check(code, "msg_maxpool2x2_bwd")
        ctx.save_for_backward(idx)
        ctx.x_shape = tuple(x_shape)
        return gx

    @staticmethod
    def backward(ctx, v):
        idx, = ctx.saved_tensors
        return _MaxPool2x2Gather.apply(v, idx, ctx.x_shape), None, None


class _MaxPool2x2Gather(Function):
    """v [B,C,H,W] at the forward's winner positions -> [B,C,H/2,W/2]; its derivative is _MaxPool2x2Backward again."""

    @staticmethod
    def forward(ctx, v, idx, x_shape):
        b, c, h, w = x_shape
        if _pitch(v) is None:
            v = v.contiguous(memory_format=torch.channels_last)
        out = torch.empty((b, c, h // 2, w // 2), dtype=v.dtype, device=v.device, memory_format=torch.channels_last)
        with _lib.on_device(v.device):
            code = _lib.lib().msg_maxpool2x2_gather(v.data_ptr(), idx.data_ptr(), out.data_ptr(), _lib.dtype_code(v), b, h, w, c,
                                                    _pitch(v), _lib.stream_of(v.device))
        _lib.check(code, "msg_maxpool2x2_gather")
        ctx.save_for_backward(idx)
        ctx.x_shape = tuple(x_shape)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        return _MaxPool2x2Backward.apply(g, idx, ctx.x_shape), None, None


def max_pool2x2(x: torch.Tensor) -> torch.Tensor:
    """F.max_pool2d(x, kernel_size=2, stride=2) for a channels-last map; shapes / layouts / types the kernel does not take
    go to the library."""
    if x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] % 2 == 0 and \
            x.shape[3] % 2 == 0 and x.shape[1] % (16 // x.element_size()) == 0 and x.shape[1] > 1 and _pitch(x) is not None:
        return _MaxPool2x2.apply(x)
    return F.max_pool2d(x, kernel_size=2, stride=2)
