"""Fused (noise +) bias + leaky-ReLU as a twice-differentiable autograd op on the gfx950 kernels.

Drop-in for the reference's op_static/fused_act.py:22-89 (``FusedLeakyReLU`` module default scale 1.0, free
function default sqrt(2)).  ``fused_bias_noise_leaky_relu`` additionally folds the generator's
``NoiseInjection`` (multi_stylegan_generator.py:288-292) into the same pass; its parameters stay where the
reference keeps them (``noise_injection.weight``, ``activation.bias``).
"""

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib
from .._autograd import _derive


def _layout(x):
    """-> (contiguous-in-its-own-layout tensor, step_b, pixels per sample, channels_last?)"""
    if x.ndim == 2:
        return x.contiguous(), 1, 1, True
    if x.ndim != 4:
        x = x.reshape(x.shape[0], x.shape[1], -1, 1)
    cl = x.shape[1] > 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    if not cl:
        x = x.contiguous()
    pix = x.shape[2] * x.shape[3]
    return x, (1 if cl else pix), pix, cl


def _noise_args(noise, x):
    if noise is None:
        return None, 1
    if noise.shape[0] not in (1, x.shape[0]) or noise.shape[1] != 1 or noise.shape[2:] != x.shape[2:]:
        raise _lib.MsgHipError(f"noise shape {tuple(noise.shape)} does not match input {tuple(x.shape)}")
    return noise.to(torch.float32).contiguous(), noise.shape[0]


def _bias_act(x, bias, ref, noise, noise_weight, grad, alpha, scale, act=3):
    """y = act(x + w*noise + b) * scale  (grad=0), or the same with the slope taken from sign(ref) (grad=1)."""
    shape = x.shape
    x, step_b, pix, _ = _layout(x)
    dev = _lib.require_gpu(x, bias, ref, noise, noise_weight)
    if ref is not None:
        ref = ref.reshape(x.shape).contiguous(memory_format=torch.channels_last if step_b == 1 and x.ndim == 4
                                             else torch.contiguous_format)
    y = torch.empty_like(x)
    if x.dtype == torch.float64:
        # the `double` of the reference's dispatch (fused_bias_act_kernel.cu:79): everything in float64
        if noise is not None:
            raise _lib.MsgHipError("the fused noise injection is an fp32 / bf16 / fp16 path; float64 takes bias only")
        b64 = None if bias is None else bias.to(torch.float64).contiguous()
        with _lib.on_device(dev):
            code = _lib.lib().msg_fused_bias_act(
                x.data_ptr(), _lib.ptr(b64), _lib.ptr(ref), y.data_ptr(), _lib.MSG_F64, x.numel(), step_b, x.shape[1],
                None, None, 1, pix, act, grad, float(alpha), float(scale), _lib.stream_of(dev))
        _lib.check(code, "msg_fused_bias_act")
        return y.reshape(shape) if y.shape != shape else y
    nz, nb = _noise_args(noise, x)
    b32 = None if bias is None else bias.to(torch.float32).contiguous()
    nw32 = None if noise_weight is None else noise_weight.to(torch.float32).contiguous()
    nbytes = (2 + (ref is not None)) * x.numel() * x.element_size()
    with _lib.on_device(dev), _lib.kernel_clock.span(('bias_act_fwd', x.dtype), nbytes):
        code = _lib.lib().msg_fused_bias_act(
            x.data_ptr(), _lib.ptr(b32), _lib.ptr(ref), y.data_ptr(), _lib.dtype_code(x, True), x.numel(), step_b,
            x.shape[1], _lib.ptr(nz), _lib.ptr(nw32), nb, pix, act, grad, float(alpha), float(scale),
            _lib.stream_of(dev))
    _lib.check(code, "msg_fused_bias_act")
    return y.reshape(shape) if y.shape != shape else y


_WS_CACHE: dict = {}     # workspace sizes of msg_bias_act_backward per problem (a pure function of the sizes)
ACT_MASK = True      # False: the activation backward always reads the stored output (tests compare both: bit-identical)


def sign_mask_for(b, c, h, w, dtype, device):
    """Buffer for the sign bytes of a [b, c, h, w] channels-last bf16 activation output ([b*h*w, c/8] uint8) that the forward
    kernels which can write them (msg_conv2d_fprop_act_mask, msg_upfirdn2d_separable_act_mask) fill beside the output:
    all the activation's backward needs of it, at a sixteenth of the size.  None when the form does not apply."""
    if not ACT_MASK or dtype != torch.bfloat16 or c % 8:
        return None
    return torch.empty((b * h * w, c // 8), dtype=torch.uint8, device=device)


def act_backward_with_head(gy, head, out_shape, noise, bias_param, need_bias, negative_slope, scale, mask):
    """The activation backward of a styled layer whose output also feeds the level's image head, with the head's data
    gradient formed inside the pass (msg_bias_act_backward_mask_head; first-order backward only):
        gpre = (gy + h) * scale * slope(mask),   h = the head's 1x1 per-sample weights applied to `head.gy`
    gy: the gradient from the layer's other consumer or None; head: conv_ops.HeadGradSlot contents (gy [B, n, H, W] bf16
    channels-last with an 8-channel pitch, base weights [n, C] fp32, style [B, C] fp32, scale).  Returns (gpre, grad_bias,
    grad_noise_weight) like FusedLeakyReLUFunctionBackward, or None when the kernel declines (the caller then takes the
    head's gradient the ordinary way)."""
    from .. import conv_ops
    b, c, h, w = out_shape
    hgy, whead, style, wscale = head
    if mask is None or hgy.dtype != torch.bfloat16 or c % 8 or hgy.shape[1] > 8 or not hgy.is_cuda:
        return None
    mbytes, tile_m, tile_n = mask
    hv, ldh = conv_ops._nhwc_view(hgy)
    if ldh != 8 or mbytes.numel() * 8 != b * c * h * w:
        return None
    dev = hgy.device
    g = None
    if gy is not None:
        if gy.dtype != torch.bfloat16 or tuple(gy.shape) != tuple(out_shape):
            return None
        g = gy if gy.is_contiguous(memory_format=torch.channels_last) else gy.contiguous(memory_format=torch.channels_last)
    gx = torch.empty((b, c, h, w), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    gb = None
    if bias_param is not None and bias_param.dtype == torch.float32 and bias_param.shape == (c,):
        gb = conv_ops._grad_dest(bias_param)
    if gb is None and need_bias:
        gb = torch.empty(c, dtype=torch.float32, device=dev)
    nz, nb = _noise_args(noise, gx)
    gnw = torch.empty(1, dtype=torch.float32, device=dev) if noise is not None else None
    need = 0
    if gb is not None or noise is not None:
        wkey = (gx.numel(), 1, c, noise is not None)
        need = _WS_CACHE.get(wkey)
        if need is None:
            need = _WS_CACHE[wkey] = _lib.lib().msg_bias_act_backward_workspace(gx.numel(), 1, c, int(noise is not None))
    ws = _lib.scratch_ptr(need, dev) if need else None
    nbytes = (1 + (g is not None)) * gx.numel() * 2 + mbytes.numel() + b * h * w * 16
    with _lib.on_device(dev), _lib.kernel_clock.span(('bias_act_bwd_mask_head', gx.dtype), nbytes):
        code = _lib.lib().msg_bias_act_backward_mask_head(
            _lib.ptr(g), hv.data_ptr(), whead.data_ptr(), style.data_ptr(), float(wscale), int(hgy.shape[1]),
            mbytes.data_ptr(), int(tile_m), int(tile_n), gx.data_ptr(), _lib.MSG_BF16, gx.numel(), c,
            _lib.ptr(gb), _lib.ptr(nz), _lib.ptr(gnw), nb, h * w, float(negative_slope), float(scale),
            ws, need, _lib.stream_of(dev))
    if code == -2:                      # MSG_EUNSUPPORTED (shapes whose workgroups would straddle samples, alignment)
        return None
    _lib.check(code, "msg_bias_act_backward_mask_head")
    return gx, (gb if gb is not None else torch.zeros(0, device=dev)), (gnw if gnw is not None else torch.zeros(0, device=dev))


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, noise, need_bias, negative_slope, scale, mask=None):
        # `need_bias` may be the bias PARAMETER itself: its slice of the flat gradient store then receives the sum directly
        # (conv_ops._grad_dest) and autograd's accumulation add of it disappears
        bias_param = need_bias if isinstance(need_bias, torch.Tensor) else None
        need_bias = True if bias_param is not None else bool(need_bias)
        # The SAVED OUTPUT's memory layout picks the kernel (channels-last vectors or planes), not the incoming gradient's:
        # a gradient that autograd summed from two consumers inherits the layout of whichever arrived first, and the engine's
        # order in second-order passes depends on thread-local node counters, i.e. on the process's history -- a
        # layout-dependent kernel choice would make the order of the bias sum, and so its last bit, vary between runs.
        if grad_output.dtype == torch.float64:
            # float64 (gradcheck): the reference's own formulation -- fused_bias_act(grad=1) and the bias sum in PyTorch
            # (op_static/fused_act.py:24-42)
            gx = _bias_act(grad_output, None, out, None, None, 1, negative_slope, scale)
            ctx.save_for_backward(out, noise)
            ctx.cfg = (negative_slope, scale)
            dims = [0] + list(range(2, gx.ndim))
            return gx, (gx.sum(dims) if need_bias else gx.new_zeros(0)), gx.new_zeros(0)
        o, step_b, pix, _ = _layout(out)
        dev = _lib.require_gpu(grad_output, o, noise)
        g = grad_output.reshape(o.shape)
        g = g.contiguous(memory_format=torch.channels_last) if (step_b == 1 and g.ndim == 4) else g.contiguous()
        gx = torch.empty_like(g)
        channels = g.shape[1]
        # grad_bias / grad_noise_weight are overwritten by a fixed-order sum of per-workgroup partials (deterministic).
        gb = None
        if bias_param is not None and bias_param.dtype == torch.float32 and bias_param.shape == (channels,):
            from ..conv_ops import _grad_dest
            gb = _grad_dest(bias_param)
        if gb is None and need_bias:
            gb = torch.empty(channels, dtype=torch.float32, device=dev)
        nz, nb = _noise_args(noise, g)
        gnw = torch.empty(1, dtype=torch.float32, device=dev) if noise is not None else None
        need = 0
        if need_bias or noise is not None:
            wkey = (g.numel(), step_b, channels, noise is not None)
            need = _WS_CACHE.get(wkey)
            if need is None:
                need = _WS_CACHE[wkey] = _lib.lib().msg_bias_act_backward_workspace(g.numel(), step_b, channels, int(noise is not None))
        ws = _lib.scratch_ptr(need, dev) if need else None        # (launch-scoped: partials -> the reduce launch of the same call)
        if mask is not None and step_b == 1 and g.ndim == 4 and g.dtype == torch.bfloat16 and channels % 8 == 0 and \
                mask[0].numel() * 8 == g.numel():
            # the forward launch left the sign bytes of `out` (bytes, tile_m, tile_n): that map is not read again
            mbytes, tile_m, tile_n = mask
            nbytes = 2 * g.numel() * g.element_size() + mbytes.numel()
            with _lib.on_device(dev), _lib.kernel_clock.span(('bias_act_bwd_mask', g.dtype), nbytes):
                code = _lib.lib().msg_bias_act_backward_mask(
                    g.data_ptr(), mbytes.data_ptr(), int(tile_m), int(tile_n), gx.data_ptr(), _lib.dtype_code(g, True),
                    g.numel(), channels,
                    _lib.ptr(gb), _lib.ptr(nz), _lib.ptr(gnw), nb, pix, float(negative_slope), float(scale),
                    ws, need, _lib.stream_of(dev))
            if code == -2:                          # MSG_EUNSUPPORTED: the backward's vector path has stricter conditions
                mask = None                         # (pointer alignment) than the forward's decision to write the bytes --
            else:                                   # the stored output is still here, take the slower path instead of raising
                _lib.check(code, "msg_bias_act_backward_mask")
        else:
            mask = None
        if mask is None:
            with _lib.on_device(dev), _lib.kernel_clock.span(('bias_act_bwd', g.dtype), 3 * g.numel() * g.element_size()):
                code = _lib.lib().msg_bias_act_backward(
                    g.data_ptr(), o.data_ptr(), gx.data_ptr(), _lib.dtype_code(g, True), g.numel(), step_b, channels,
                    _lib.ptr(gb), _lib.ptr(nz), _lib.ptr(gnw), nb, pix, float(negative_slope), float(scale),
                    ws, need, _lib.stream_of(dev))
            _lib.check(code, "msg_bias_act_backward")
        ctx.save_for_backward(out, noise)
        ctx.cfg = (negative_slope, scale)
        ctx.mask = mask                  # (sign bytes, when this pass used them: its own backward is the same multiplication)
        gx = gx.reshape(grad_output.shape)
        if gb is None:
            gb = torch.zeros(0, device=dev)
        if gnw is None:
            gnw = torch.zeros(0, device=dev)
        return gx, gb, gnw

    @staticmethod
    def backward(ctx, gg_input, gg_bias, gg_noise_weight):
        out, noise = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        if gg_input is None:
            gg_input = torch.zeros_like(out)
        ggb = gg_bias if (gg_bias is not None and gg_bias.numel()) else None
        ggw = gg_noise_weight if (noise is not None and gg_noise_weight is not None
                                  and gg_noise_weight.numel()) else None
        # linear in (gg_input, gg_bias, gg_noise_weight); the mask has zero derivative (reference fused_act.py:45-51)
        mask = getattr(ctx, "mask", None)
        if ggb is None and ggw is None and mask is not None and gg_input.dtype == torch.bfloat16 and gg_input.ndim == 4 and \
                gg_input.shape == out.shape and gg_input.shape[1] % 8 == 0 and mask[0].numel() * 8 == gg_input.numel():
            # the cotangent times the slope, from the sign bytes (2 1/16 maps moved instead of 3: the regularisers' second
            # backward runs this once per activation)
            g = gg_input if gg_input.is_contiguous(memory_format=torch.channels_last) else \
                gg_input.contiguous(memory_format=torch.channels_last)
            gx = torch.empty_like(g)
            dev = g.device
            mbytes, tile_m, tile_n = mask
            with _lib.on_device(dev), _lib.kernel_clock.span(('bias_act_bwd_mask', g.dtype), 2 * g.numel() * 2 + mbytes.numel()):
                code = _lib.lib().msg_bias_act_backward_mask(
                    g.data_ptr(), mbytes.data_ptr(), int(tile_m), int(tile_n), gx.data_ptr(), _lib.MSG_BF16, g.numel(),
                    g.shape[1], None, None, None, 1, g.shape[2] * g.shape[3], float(negative_slope), float(scale), None, 0,
                    _lib.stream_of(dev))
            if code != -2:
                _lib.check(code, "msg_bias_act_backward_mask")
                return gx, None, None, None, None, None, None
        gg_out = _bias_act(gg_input, ggb, out, noise if ggw is not None else None, ggw, 1, negative_slope, scale)
        return gg_out, None, None, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, x, bias, noise, noise_weight, negative_slope, scale):
        out = _bias_act(x, bias, None, noise, noise_weight, 0, negative_slope, scale)
        ctx.save_for_backward(out, noise)
        ctx.cfg = (negative_slope, scale, bias is not None, noise_weight is not None)
        ctx.bias_param = bias
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, noise = ctx.saved_tensors
        negative_slope, scale, has_bias, has_nw = ctx.cfg
        gx, gb, gnw = _derive(FusedLeakyReLUFunctionBackward, grad_output, out, noise if has_nw else None,
                                                           ctx.bias_param if has_bias else False, negative_slope, scale)
        return gx, (gb if has_bias else None), None, (gnw if has_nw else None), None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, None, None, negative_slope, scale)


def fused_bias_noise_leaky_relu(input, bias, noise, noise_weight, negative_slope=0.2, scale=1.0):
    """lrelu(input + noise_weight * noise + bias) * scale in one pass; noise is [B or 1, 1, H, W] (no grad)."""
    return FusedLeakyReLUFunction.apply(input, bias, noise, noise_weight, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=1.):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


class _ScaledAdd(Function):
    """y = (a + b) * gain in one pass (csrc/bias_act.hip: msg_scaled_add); the backward is plain torch (gy * gain for
    both inputs) so that it stays differentiable for R1."""

    @staticmethod
    def forward(ctx, a, b, gain):
        ctx.gain = gain
        dev = _lib.require_gpu(a, b)
        y = torch.empty_like(a)
        with _lib.on_device(dev):
            code = _lib.lib().msg_scaled_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.dtype_code(a), a.numel(), 1.0,
                                             float(gain), _lib.stream_of(dev))
        _lib.check(code, "msg_scaled_add")
        return y

    @staticmethod
    def backward(ctx, gy):
        g = gy * ctx.gain
        return g, g, None


class _GammaMerge(Function):
    """y = (gamma * a + b) * gain, gamma a 0-d fp32 parameter (csrc/bias_act.hip: msg_gamma_merge), one launch forward and one
    pass backward; a second-order graph (R1) is built from the torch formulation."""

    @staticmethod
    def forward(ctx, a, b, gamma, gain):
        dev = _lib.require_gpu(a, b, gamma)
        y = torch.empty_like(a)
        g32 = gamma.detach().reshape(1).to(torch.float32)
        with _lib.on_device(dev):
            code = _lib.lib().msg_gamma_merge(a.data_ptr(), b.data_ptr(), g32.data_ptr(), y.data_ptr(), _lib.dtype_code(a),
                                              a.numel(), float(gain), _lib.stream_of(dev))
        _lib.check(code, "msg_gamma_merge")
        ctx.gain = float(gain)
        ctx.save_for_backward(a, gamma)
        return y

    @staticmethod
    def backward(ctx, gy):
        a, gamma = ctx.saved_tensors
        gain = ctx.gain
        if torch.is_grad_enabled() or not (gy.stride() == a.stride() and gy.dtype == a.dtype):
            from ..conv_ops import _consumed
            g = gy * gain
            # (gamma's gradient only where the engine will use it: the first pass of R1 differentiates with respect to the
            #  images alone, and <gy, a> in fp32 was two casts, a product and a reduction over the whole map for nothing)
            g_gamma = (gy.float() * a.float()).sum().to(gamma.dtype).reshape(gamma.shape) * gain if _consumed(ctx, 2, 2) else None
            return g * gamma.to(g.dtype), g, g_gamma, None
        dev = gy.device
        ga, gb = torch.empty_like(a), torch.empty_like(a)
        gg = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib().msg_gamma_merge_backward_workspace(), dtype=torch.float32, device=dev)
        g32 = gamma.detach().reshape(1).to(torch.float32)
        with _lib.on_device(dev):
            code = _lib.lib().msg_gamma_merge_backward(gy.data_ptr(), a.data_ptr(), g32.data_ptr(), ga.data_ptr(), gb.data_ptr(),
                                                       gg.data_ptr(), _lib.dtype_code(a), a.numel(), gain, ws.data_ptr(),
                                                       _lib.stream_of(dev))
        _lib.check(code, "msg_gamma_merge_backward")
        return ga, gb, gg.reshape(gamma.shape).to(gamma.dtype), None


def gamma_merge(a, b, gamma, gain):
    """(gamma * a + b) * gain for two maps of identical shape, dtype and memory layout and a 0-d parameter gamma
    (the NonLocalBlock's merge, u_net_2d_discriminator.py:381); other operands take the torch formulation."""
    same = a.shape == b.shape and a.dtype == b.dtype and a.stride() == b.stride() and a.is_cuda and gamma.is_cuda and \
        gamma.numel() == 1 and a.numel() % 8 == 0 and a.dtype in (torch.float32, torch.bfloat16) and \
        (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last))
    if not same:
        return scaled_add(gamma.to(a.dtype) * a, b, gain)
    return _GammaMerge.apply(a, b, gamma, gain)


def _rows_view(t):
    """(rows, cols, pitch) if `t` [B,C,H,W] is a channels-last map or a channel-slice of one, else None."""
    if t.ndim != 4 or not t.is_cuda:
        return None
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    if (sc == 1 or c == 1) and sh == w * sw and (sb == h * w * sw or b == 1) and sw >= c:
        return b * h * w, c, sw
    return None


class _ScaledAddRows(Function):
    """(a + b) * gain for channels-last maps / channel-slices with different pitches (csrc: msg_scaled_add_rows)."""

    @staticmethod
    def forward(ctx, a, b, gain):
        ctx.gain = gain
        dev = _lib.require_gpu(a, b)
        (rows, cols, lda), (_, _, ldb) = _rows_view(a), _rows_view(b)
        y = torch.empty(a.shape, dtype=a.dtype, device=dev, memory_format=torch.channels_last)
        with _lib.on_device(dev):
            code = _lib.lib().msg_scaled_add_rows(a.data_ptr(), b.data_ptr(), y.data_ptr(), _lib.dtype_code(a), rows, cols,
                                                  lda, ldb, cols, 1.0, float(gain), _lib.stream_of(dev))
        _lib.check(code, "msg_scaled_add_rows")
        return y

    @staticmethod
    def backward(ctx, gy):
        g = gy * ctx.gain
        return g, g, None


def _rows_ok(a, b):
    if a.shape != b.shape or a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16):
        return False
    va, vb = _rows_view(a), _rows_view(b)
    vec = 16 // a.element_size()
    return va is not None and vb is not None and va[1] % vec == 0 and va[2] % vec == 0 and vb[2] % vec == 0 and \
        a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and a.shape[1] > 1


class _ScaledAddFork(Function):
    """(a + b) * gain handed out TWICE (two aliases of one result), for a tensor with exactly two consumers: autograd
    then delivers the two incoming gradients to THIS node separately, and the backward is ONE pass (g1 + g2) * gain
    (msg_scaled_add again) instead of autograd's accumulation add followed by the gain multiply."""

    @staticmethod
    def forward(ctx, a, b, gain):
        ctx.gain = gain
        y = _ScaledAdd.forward(ctx, a, b, gain)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None or g2 is None:
            g = (g1 if g1 is not None else g2) * ctx.gain
        elif _rows_ok(g1, g2):
            g = _derive(_ScaledAddRows, g1, g2, ctx.gain)  # differentiable (R1 runs a second-order pass through here)
        else:
            g = (g1 + g2) * ctx.gain
        return g, g, None


def scaled_add_fork(a, b, gain):
    """scaled_add whose result feeds two consumers: returns two aliases (see _ScaledAddFork)."""
    same = a.shape == b.shape and a.dtype == b.dtype and a.stride() == b.stride() and a.is_cuda and \
        a.numel() % 8 == 0 and a.dtype in (torch.float32, torch.bfloat16) and \
        (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last))
    if not same:
        y = (a + b) * gain
        return y, y
    return _ScaledAddFork.apply(a, b, gain)


def scaled_add(a, b, gain):
    """(a + b) * gain for two tensors of identical shape, dtype and memory layout (falls back to torch otherwise)."""
    same = a.shape == b.shape and a.dtype == b.dtype and a.stride() == b.stride() and a.is_cuda and \
        a.numel() % 8 == 0 and a.dtype in (torch.float32, torch.bfloat16) and \
        (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last))
    return _ScaledAdd.apply(a, b, gain) if same else (a + b) * gain
