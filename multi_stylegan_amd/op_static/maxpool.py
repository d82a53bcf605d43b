"""2x2 / stride-2 max-pooling of channels-last maps on the gfx950 kernels (csrc/maxpool.hip) -- the pooling of the NonLocalBlock's
key and value maps (reference u_net_2d_discriminator.py:366-370: F.max_pool2d(phi(x)), F.max_pool2d(g(x)))."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

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
        if torch.is_grad_enabled():
            # a second-order graph is being built (R1): the library's differentiable pooling on the saved input
            with torch.enable_grad():
                xx = x if x.requires_grad else x.detach().requires_grad_(True)
                return torch.autograd.grad(F.max_pool2d(xx, kernel_size=2, stride=2), xx, gy, create_graph=True)[0]
        b, c, h, w = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((b, c, h, w), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        with _lib.on_device(gy.device):
            code = _lib.lib().msg_maxpool2x2_bwd(gy.data_ptr(), idx.data_ptr(), gx.data_ptr(), _lib.dtype_code(gy), b, h, w, c,
                                                 _lib.stream_of(gy.device))
        _lib.check(code, "msg_maxpool2x2_bwd")
        return gx


def max_pool2x2(x: torch.Tensor) -> torch.Tensor:
    """F.max_pool2d(x, kernel_size=2, stride=2) for a channels-last map; shapes / layouts / types the kernel does not take
    go to the library."""
    if x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] % 2 == 0 and \
            x.shape[3] % 2 == 0 and x.shape[1] % (16 // x.element_size()) == 0 and x.shape[1] > 1 and _pitch(x) is not None:
        return _MaxPool2x2.apply(x)
    return F.max_pool2d(x, kernel_size=2, stride=2)
