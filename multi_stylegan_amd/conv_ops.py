"""Dense contractions of the hot path on the hand-written MFMA kernels (csrc/conv_fprop_pp.hip, conv_fprop.hip,
conv_wgrad.hip), the modulation arithmetic around them (csrc/modulate.hip, relayout.hip) and the few-row linear
layers (csrc/linear.hip).

Three primitives per convolution geometry, closed under differentiation so that first AND second order autograd
(R1 on the discriminator, path-length regularisation on the generator) run on the same two kernels:

    F(x, w)  -> y        forward contraction
    D(gy, w) -> gx       data gradient        (same kernel as F with the weights re-laid)
    G(gy, x) -> gw       weight gradient      (TN kernel)

    dF = (D(gy, w), G(gy, x));   dD = (F(v, w), G(gy, v));   dG = (F(x, u), D(gy, u))

Geometries: "conv" (kh x kw, stride 1 or 2, any padding) and "up2" (the generator's 2x2 stride-2 transposed conv,
run as a 1x1 contraction to 4*O channels stored pixel-shuffled).  Weights are given in the reference's parameter
layout ([O,I,kh,kw], or [B,O,I,kh,kw] for the per-sample weights of the modulated convolution); their K-contiguous
kernel-side images are built by one kernel per parameter and cached until that parameter's optimizer steps
(_param_images); activations are channels-last with a 16-byte-aligned channel stride.  Fused forms: conv + activation
(_ConvActF, _ModulatedConv with fuse_act), conv + residual merge (_ConvResidualF).
There is no CPU or library fallback.
"""
import math
import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib
from ._autograd import _derive


# ----------------------------------------------------------------------------------------------- layout helpers
def _vec(dtype) -> int:
    return 8 if dtype == torch.bfloat16 else 4


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _padded_nhwc(b, c, h, w, dtype, device):
    """Logical [b, c, h, w] view of a fresh NHWC buffer whose channel stride is a whole number of 16-byte vectors (pad
    channels zeroed): what every kernel launch accepts as is."""
    cx = _round_up(c, _vec(dtype))
    buf = torch.empty((b, h, w, cx), dtype=dtype, device=device)
    if cx != c:
        buf[..., c:].zero_()
    return buf.permute(0, 3, 1, 2)[:, :c]


class _CatChannels(Function):
    """Channel concatenation written straight into a padded channels-last buffer (one strided copy per piece, no
    re-padding at the launches that read it); the backward hands out channel-slice VIEWS of the incoming gradient."""

    @staticmethod
    def forward(ctx, *tensors):
        t0 = tensors[0]
        ctx.splits = [t.shape[1] for t in tensors]
        out = _padded_nhwc(t0.shape[0], sum(ctx.splits), t0.shape[2], t0.shape[3], t0.dtype, t0.device)
        off = 0
        for t in tensors:
            out[:, off:off + t.shape[1]].copy_(t)
            off += t.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        if torch.is_grad_enabled():                    # second-order pass (R1): keep the split's own backward a concat
            pieces = _SplitChannels.apply(g, *ctx.splits)
            return tuple(p if need else None for p, need in zip(pieces, ctx.needs_input_grad))
        outs, off = [], 0
        for i, c in enumerate(ctx.splits):
            outs.append(g[:, off:off + c] if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(outs)


class _SplitChannels(Function):
    """Channel-slice views of a gradient whose own backward is ONE concatenation (autograd's slice backward would
    allocate a zero map per piece, copy the piece in and add the maps up)."""

    @staticmethod
    def forward(ctx, g, *splits):
        ctx.splits = splits
        ctx.meta = (g.shape[0], g.shape[2], g.shape[3], g.dtype, g.device)
        outs, off = [], 0
        for c in splits:
            outs.append(g[:, off:off + c])
            off += c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gg):
        b, h, w, dtype, dev = ctx.meta
        pieces = [x if x is not None else torch.zeros((b, c, h, w), dtype=dtype, device=dev)
                  for x, c in zip(gg, ctx.splits)]
        return (_CatChannels.apply(*pieces),) + (None,) * len(ctx.splits)


def to_compute_layout(x: torch.Tensor, dtype=None) -> torch.Tensor:
    """Feature maps are kept channels-last (NHWC in HBM, NCHW logical shape).  Channel counts that are not a whole
    number of 16-byte vectors (the 6-channel image pair) are padded HERE, once, instead of at every launch that reads
    the tensor (forward, weight gradient, second-order terms)."""
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    if x.ndim == 4 and x.shape[1] > 1:
        if x.is_cuda and x.shape[1] % _vec(x.dtype):
            return _CatChannels.apply(x)
        return x.contiguous(memory_format=torch.channels_last)
    return x.contiguous()


class _CatAliased(Function):
    """torch.cat(pieces, dim=1) into `buffer` (cat_destination) for pieces that their producers WROTE into their
    channel-slices of it: those cost nothing here, any other piece is copied into its slice; the result is the buffer.
    The backward hands out channel-slice views of the incoming gradient exactly as _CatChannels does."""

    @staticmethod
    def forward(ctx, buffer, *pieces):
        ctx.splits = [t.shape[1] for t in pieces]
        off = 0
        for t in pieces:
            want = buffer[:, off:off + t.shape[1]]
            if t.shape != want.shape or t.dtype != buffer.dtype:
                raise _lib.MsgHipError(f"cat_in_place: piece {tuple(t.shape)} for slice {tuple(want.shape)}")
            same = t.data_ptr() == want.data_ptr() and \
                all(a == b for a, b, n in zip(t.stride(), want.stride(), t.shape) if n != 1)    # (size-1 dims: any stride)
            if not same:
                if t.untyped_storage().data_ptr() == buffer.untyped_storage().data_ptr():
                    raise _lib.MsgHipError("cat_in_place: a piece lives in the destination buffer, but not in its slice")
                want.copy_(t)                          # (not produced in place: one strided copy, as _CatChannels)
            off += t.shape[1]
        if off != buffer.shape[1]:
            raise _lib.MsgHipError("cat_in_place: the pieces do not fill the destination buffer")
        return buffer.view_as(buffer)

    @staticmethod
    def backward(ctx, g):
        needs = ctx.needs_input_grad[1:]
        if torch.is_grad_enabled():                    # second-order pass (R1): keep the split's own backward a concat
            pieces = _SplitChannels.apply(g, *ctx.splits)
            return (None, *(p if need else None for p, need in zip(pieces, needs)))
        out, off = [], 0
        for c, need in zip(ctx.splits, needs):
            out.append(g[:, off:off + c] if need else None)
            off += c
        return (None, *out)


def cat_destination(b: int, channels, h: int, w: int, dtype, device):
    """Buffer for a channel concatenation whose pieces are written in place by their producers, and its slices:
    -> (buffer [b, sum(channels), h, w] channels-last, [slice views]), or None when a piece is not a whole number of
    16-byte channel vectors (then cat_channels copies, as before)."""
    if any(c % _vec(dtype) for c in channels) or not torch.device(device).type == "cuda":
        return None
    buf = _padded_nhwc(b, sum(channels), h, w, dtype, device)
    if any(c % (128 // buf.element_size()) for c in channels):
        # The contraction kernels read whole 128-byte channel runs and rely on zero WEIGHTS beyond a layer's real input
        # channels: a slice that is not a whole number of runs is read together with its neighbour, which may not have
        # been written yet -- and uninitialised memory can hold NaN bit patterns.  (None of the models' levels: 128 ... 768.)
        buf.zero_()
    views, off = [], 0
    for c in channels:
        views.append(buf[:, off:off + c])
        off += c
    return buf, views


def cat_in_place(buffer, pieces) -> torch.Tensor:
    """The concatenation whose pieces already live in `buffer` (see cat_destination)."""
    return _CatAliased.apply(buffer, *pieces)


def cat_channels(tensors) -> torch.Tensor:
    """torch.cat(tensors, dim=1) in the compute layout."""
    tensors = list(tensors)
    if tensors[0].is_cuda and all(t.dtype == tensors[0].dtype for t in tensors):
        return _CatChannels.apply(*tensors)
    return to_compute_layout(torch.cat(tensors, dim=1))


def _nhwc_view(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """-> (tensor whose memory is [B, H, W, Cx] with the real channels first, Cx).  Accepts channels-last tensors and
    channel-slices of channels-last buffers as they are; anything else is copied into a zero-padded NHWC buffer."""
    b, c, h, w = x.shape
    vec = _vec(x.dtype)
    sb, sc, sh, sw = x.stride()
    cx = sw
    if (sc == 1 or c == 1) and cx >= c and cx % vec == 0 and sh == w * cx and (sb == h * w * cx or b == 1) \
            and x.data_ptr() % 16 == 0:
        return x, cx
    cx = _round_up(c, vec)
    # one pass: a channels-last buffer by construction, the copy, and zeros only in the padding channels (this used to
    # be zeros -> .contiguous(channels_last) -> copy_: three passes over maps of up to a gigabyte in the second-order graphs)
    buf = torch.empty((b, h, w, cx), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    if cx == c:
        buf.copy_(x)
        return buf, cx
    buf[:, :c].copy_(x)
    buf[:, c:].zero_()
    return buf[:, :c], cx


def _alloc_out(b: int, n: int, h: int, w: int, dtype, device) -> Tuple[torch.Tensor, int]:
    """Channels-last output with the channel stride rounded up to 16 bytes; returns (logical [b,n,h,w] view, ldy)."""
    ld = _round_up(n, _vec(dtype))
    if ld == n:                        # (the common case in ONE torch call: same strides as the permuted buffer below)
        return torch.empty((b, n, h, w), dtype=dtype, device=device, memory_format=torch.channels_last), ld
    buf = torch.empty((b, h, w, ld), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :n], ld


def _relayout_kernel_ok(w, dtype, taps) -> bool:
    return w.is_cuda and w.dtype == torch.float32 and dtype in (torch.bfloat16, torch.float32) and \
        taps <= 16 and w.ndim >= 4


def _relay_fwd(w: torch.Tensor, dtype) -> Tuple[torch.Tensor, int]:
    """[..., O, I, kh, kw] -> [..., O, kh*kw, Ck] (K-contiguous, input channels zero-padded to a 128-byte run)."""
    if w.ndim == 2:
        w = w[:, :, None, None]
    *lead, o, i, kh, kw = w.shape
    ck = _round_up(i, 128 // (2 if dtype == torch.bfloat16 else 4))
    if _relayout_kernel_ok(w, dtype, kh * kw):
        # per-sample (or any non-parameter) fp32 weights: the kernel of _param_images, the leading dimensions folded
        # into the rows -- one pass instead of a zero fill and a transposing, casting copy
        rows = w.numel() // (i * kh * kw)
        out = torch.empty((*lead, o, kh * kw, ck), dtype=dtype, device=w.device)
        with _lib.on_device(w.device):
            code = _lib.lib().msg_relayout_weight(w.detach().contiguous().data_ptr(), out.data_ptr(), None, None,
                                                  _lib.dtype_code(out), rows, i, kh * kw, ck, ck, 0, 0, 1.0,
                                                  _lib.stream_of(w.device))
        _lib.check(code, "msg_relayout_weight")
        return out, ck
    out = torch.zeros((*lead, o, kh * kw, ck), dtype=dtype, device=w.device)
    out[..., :i] = w.reshape(*lead, o, i, kh * kw).transpose(-1, -2)
    return out, ck


def _relay_dgrad(w: torch.Tensor, dtype, flip: bool) -> Tuple[torch.Tensor, int]:
    """[..., O, I, kh, kw] -> [..., I, kh*kw (flipped if asked), Ok]: the weights of the data-gradient contraction."""
    if w.ndim == 2:
        w = w[:, :, None, None]
    *lead, o, i, kh, kw = w.shape
    ok = _round_up(o, 128 // (2 if dtype == torch.bfloat16 else 4))
    if _relayout_kernel_ok(w, dtype, kh * kw) and len(lead) <= 1:
        n = lead[0] if lead else 1
        out = torch.empty((*lead, i, kh * kw, ok), dtype=dtype, device=w.device)
        w3 = w.detach().contiguous().view(n, o, i, kh * kw)
        ov = out.view(n, i, kh * kw, ok)
        with _lib.on_device(w.device):
            for k in range(n):                      # (the data-gradient image is per sample: one small launch each)
                code = _lib.lib().msg_relayout_weight(w3[k].data_ptr(), None, ov[k].data_ptr(), None, _lib.dtype_code(out),
                                                      o, i, kh * kw, ok, ok, int(flip), 0, 1.0, _lib.stream_of(w.device))
                _lib.check(code, "msg_relayout_weight")
        return out, ok
    src = w.reshape(*lead, o, i, kh * kw)
    if flip:
        src = src.flip(-1)
    out = torch.zeros((*lead, i, kh * kw, ok), dtype=dtype, device=w.device)
    out[..., :o] = src.permute(*range(len(lead)), len(lead) + 1, len(lead) + 2, len(lead))
    return out, ok


def _s2_plan(k: int, pad: int):
    """Stride-2 data gradient as a stride-1 gather over gy, one sub-filter per output parity (1-D plan).
    gx[2c + ph] = sum_u gy[c + e - u] * w[t0 + 2u]  with t0 = (ph + pad) % 2, e = (ph + pad - t0) / 2.
    -> (taps U, pad', table[ph][tau] = forward tap index or -1): gy index = c + tau - pad'."""
    offs = {}
    for ph in (0, 1):
        t0 = (ph + pad) % 2
        e = (ph + pad - t0) // 2
        for u in range((k - t0 + 1) // 2):
            offs[(ph, e - u)] = t0 + 2 * u
    dmin = min(d for _, d in offs)
    dmax = max(d for _, d in offs)
    taps = dmax - dmin + 1
    table = [[offs.get((ph, tau + dmin), -1) for tau in range(taps)] for ph in (0, 1)]
    return taps, -dmin, table


def _relay_dgrad_s2(w: torch.Tensor, dtype, pad: int) -> Tuple[torch.Tensor, int, int, int]:
    """[..., O, I, k, k] -> [..., 4*I, U*U, Ok]: the stride-2 data gradient as ONE stride-1 conv over gy with U x U taps
    whose 4*I output channels are the four output parities (written pixel-shuffled by the kernel).  Compared with
    the zero-insertion form (in_up = 2) the matrix cores skip the 3/4 of the taps that only ever meet parity holes."""
    *lead, o, i, kh, kw = w.shape
    assert kh == kw
    taps, pad2, table = _s2_plan(kh, pad)
    ok = _round_up(o, 128 // (2 if dtype == torch.bfloat16 else 4))
    # one gather instead of a copy per (parity, tap): source tap index per (parity, tap), kh * kw = "no tap" (a zero plane)
    src = [[(table[ph][th] * kw + table[pw][tw]) if table[ph][th] >= 0 and table[pw][tw] >= 0 else kh * kw
            for th in range(taps) for tw in range(taps)] for ph in (0, 1) for pw in (0, 1)]
    idx = torch.tensor(src, dtype=torch.long, device=w.device)                   # [4, taps^2]
    wt = w.transpose(-4, -3).reshape(*lead, i, o, kh * kw)                       # [..., I, O, kh*kw]
    wz = torch.cat([wt, wt.new_zeros((*lead, i, o, 1))], dim=-1)                 # (+ the zero plane)
    picked = wz[..., idx]                                                        # [..., I, O, 4, taps^2]
    n = len(lead)
    picked = picked.permute(*range(n), n + 2, n, n + 3, n + 1)                   # [..., 4, I, taps^2, O]
    out = torch.zeros((*lead, 4, i, taps * taps, ok), dtype=dtype, device=w.device)
    out[..., :o] = picked
    return out.reshape(*lead, 4 * i, taps * taps, ok), ok, taps, pad2


# Re-laid copies of PARAMETERS are cached between weight updates: D runs three times and G two to three times per
# iteration on unchanged weights.  The cache lives ON the nn.Parameter object (it dies with it and can never alias a
# recycled address) and is validated against (a) the tensor's in-place version counter and (b) a global weight
# generation that every torch optimizer step bumps (fused/foreach optimizers and `.data` writes do not touch the
# version counter).  Code that writes parameters behind both mechanisms must call invalidate_weight_cache().
_WEIGHT_GENERATION = [0]


def invalidate_weight_cache(params=None) -> None:
    """Declare parameters changed behind autograd's back (`.data` writes, fused / foreach optimizers do not bump the
    tensors' version counters): the given ones, or -- with no argument -- every cached image of every parameter."""
    if params is None:
        _WEIGHT_GENERATION[0] += 1
        return
    for p in params:
        p.__dict__["_msg_gen"] = p.__dict__.get("_msg_gen", 0) + 1


def _after_optimizer_step(optimizer, *_args, **_kwargs) -> None:
    # only the parameters THIS optimizer owns changed: the discriminator's step must not throw away the generator's
    # re-laid weights (and vice versa) -- with one global generation every weight was re-laid twice per iteration
    for group in optimizer.param_groups:
        invalidate_weight_cache(group["params"])


def _stamp(w):
    return (w._version, _WEIGHT_GENERATION[0], w.__dict__.get("_msg_gen", 0), w.data_ptr())


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_optimizer_step)


def _cached(w, tag, dtype, wscale, build):
    if not isinstance(w, torch.nn.Parameter):
        out = build()
        return (out[0] * wscale, *out[1:]) if wscale != 1.0 else out
    store = w.__dict__.setdefault("_msg_relay", {})
    key = (tag, dtype, wscale)
    stamp = _stamp(w)
    hit = store.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        out = build()
        if wscale != 1.0:
            out = (out[0] * wscale, *out[1:])
    store[key] = (stamp, out)
    return out


def _param_images(w, dtype, gain, kind, modulation=False):
    """{'f': (fwd image, ck), 'd': (dgrad image, ok)[, 'wsq': sum_t w^2]} of a shared-weight PARAMETER, built by ONE
    kernel (csrc/relayout.hip) per weight update and cached on the parameter.  `modulation`: the fp32, unpadded base
    images + wsq that the modulated conv scales per sample.  Returns None when the fast path does not apply (then the
    torch re-layout functions are used)."""
    key = ("img", dtype, gain, kind, modulation)
    store = w.__dict__.get("_msg_relay")
    if store is not None:                      # (the hit path first: ~240 lookups per training iteration on the host's critical path)
        hit = store.get(key)
        if hit is not None and hit[0] == (w._version, _WEIGHT_GENERATION[0], w.__dict__.get("_msg_gen", 0), w.data_ptr()):
            return hit[1]
    if not (isinstance(w, torch.nn.Parameter) and w.is_cuda and w.dtype == torch.float32):
        return None
    o, i = _oi(w)
    t = w.numel() // (o * i)
    if t > 16 or w.numel() != o * i * t:
        return None
    store = w.__dict__.setdefault("_msg_relay", {})
    stamp = _stamp(w)
    dev = w.device
    unit = 1 if modulation else 128 // (2 if dtype == torch.bfloat16 else 4)
    ck, ok = _round_up(i, unit), _round_up(o, unit)
    up2 = kind == "up2"
    with torch.no_grad():
        w3 = w.detach().reshape(o, i, t).contiguous()
        fwd = torch.empty((t * o, 1, ck) if up2 else (o, t, ck), dtype=dtype, device=dev)
        dgr = torch.empty((i, t, ok), dtype=dtype, device=dev)
        wsq = torch.empty((o, i), dtype=torch.float32, device=dev) if modulation else None
        with _lib.on_device(dev):
            code = _lib.lib().msg_relayout_weight(w3.data_ptr(), fwd.data_ptr(), dgr.data_ptr(), _lib.ptr(wsq),
                                                  _lib.dtype_code(fwd), o, i, t, ck, ok, int(not up2), int(up2),
                                                  float(gain), _lib.stream_of(dev))
        _lib.check(code, "msg_relayout_weight")
    out = {"f": (fwd, ck), "d": (dgr, ok), "wsq": wsq}
    store[key] = (stamp, out)
    return out


# --------------------------------------------------------------------------------------------------- raw launches
_CLOCK_SHAPES = bool(int(os.environ.get("MSG_CLOCK_SHAPES", "0")))   # per-shape timing keys (tools/shape_table.py)

# How the contractions of the fp32-STORAGE path multiply (conv forward / data gradient / weight gradient; include/msg_hip.h):
#   "exact"        -- v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain (the default: what the 1e-3 parity gate was built on);
#   "split_bf16x3" -- MSG_F32_SPLIT: every product as six bf16 MFMA products on (hi, mid, lo) splits of the operands (all 24
#                     mantissa bits), fp32 accumulation: fp32-rounding-level error, held to the SAME step-trace tolerances
#                     (tests/test_hip_models.py); bench.py reports it as its own leg.
FP32_CONTRACTION = "exact"
MSG_F32_SPLIT = 4
_SPLIT_CODES = {"split_bf16x3": MSG_F32_SPLIT}


class fp32_contraction:
    """``with conv_ops.fp32_contraction("split_bf16x3"): ...`` (or call ``.set`` for good)."""

    def __init__(self, mode: str):
        if mode not in ("exact", "split_bf16x3"):
            raise ValueError(f"fp32 contraction mode {mode!r}: 'exact' or 'split_bf16x3'")
        self.mode, self.prev = mode, None

    def __enter__(self):
        global FP32_CONTRACTION
        self.prev, FP32_CONTRACTION = FP32_CONTRACTION, self.mode
        return self

    def __exit__(self, *exc):
        global FP32_CONTRACTION
        FP32_CONTRACTION = self.prev
        return False

    @staticmethod
    def set(mode: str) -> None:
        fp32_contraction(mode).__enter__()


def _contraction_code(t: torch.Tensor, mode: Optional[str] = None) -> int:
    """Storage code of a contraction operand: fp32 maps carry MSG_F32_SPLIT when the split-bf16 products are selected.
    ``mode``: the contraction mode captured when the layer's FORWARD ran (Geometry.mode) -- its backward and double backward
    multiply the same way even when they run after a ``with fp32_contraction(...)`` block has ended; None = the current one."""
    code = _lib.dtype_code(t)
    mode = FP32_CONTRACTION if mode is None else mode
    return _SPLIT_CODES[mode] if (code == _lib.MSG_F32 and mode != "exact") else code


_PLAN_CACHE: dict = {}
_WGRAD_WS_CACHE: dict = {}


def _launch_fprop(x, wk, ck, bias, n, out_hw, kh, kw, stride, pad, in_up, pixel_shuffle, per_sample, c_real,
                  flops=None, act=None, residual=None, out=None, mode=None):
    """act = (act_bias | None, noise | None, noise_weight | None, alpha, scale): fuse the layer's activation stage.
    residual = (map shaped like the output, gain): y = (conv + map) * gain in the epilogue.
    out: a channels-last map or channel-slice of a wider one that receives the result (cat_destination)."""
    dev = _lib.require_gpu(x, wk, bias)
    xv, cx = _nhwc_view(x)
    b, _, ih, iw = xv.shape
    oh, ow = out_hw
    if out is not None:
        y, ldy = _nhwc_view(out)
        if y is not out or pixel_shuffle or out.shape != (b, n, oh, ow) or out.dtype != x.dtype or n % _vec(x.dtype):
            raise _lib.MsgHipError(f"conv fprop: destination {tuple(out.shape)} / {out.stride()} for a {(b, n, oh, ow)} result")
    elif pixel_shuffle:
        y, ldy = _alloc_out(b, n // 4, 2 * oh, 2 * ow, x.dtype, dev)
    else:
        y, ldy = _alloc_out(b, n, oh, ow, x.dtype, dev)
    # (host-side shape check in front of the launch: a weight image with extra leading dimensions would give the kernel a
    #  wrong per-sample stride and send it out of bounds)
    rows_needed = n * (1 if pixel_shuffle else kh * kw) * ck
    if per_sample:
        # (one sample's image must be dense: checked on the strides -- `wk[0].is_contiguous()` built a view tensor per launch)
        ws_ = wk.stride() if wk.ndim == 4 else None
        if ws_ is None or wk.shape[0] != b or ws_[0] < rows_needed or ws_[3] != 1 or \
                (wk.shape[2] > 1 and ws_[2] != wk.shape[3]) or (wk.shape[1] > 1 and ws_[1] != wk.shape[2] * wk.shape[3]):
            raise _lib.MsgHipError(f"conv fprop: per-sample weight image {tuple(wk.shape)} for batch {b}, {n} x {kh * kw} x {ck}")
    elif wk.numel() < rows_needed or not wk.is_contiguous():
        raise _lib.MsgHipError(f"conv fprop: weight image {tuple(wk.shape)} smaller than {n} x {kh * kw} x {ck}")
    wstride = wk.stride(0) if per_sample else 0
    # algorithmic FLOPs: real channels, and only the taps a transposed strided conv can reach (1/in_up^2)
    if flops is None:
        flops = 2.0 * b * oh * ow * n * kh * kw * c_real / (in_up * in_up)
    key = "conv_fprop"
    if _lib.kernel_clock.enabled:                       # label the timing with the kernel the library will pick
        plan = _lib.lib().msg_conv2d_fprop_plan(_lib.dtype_code(x), b, ih, iw, cx, ck, oh, ow, n, kh, kw, wstride)
        key = ("conv_fprop_reg", "conv_fprop_dma", "conv_fprop_pp", "conv_fprop_row3", "conv_fprop_row3n",
               "conv_fprop_thin")[plan]
        if plan == 5 and act is not None:
            key = "conv_fprop_reg"
        if x.dtype == torch.bfloat16 and bias is None and act is None and residual is None and \
                _lib.lib().msg_conv2d_fprop_upconv_eligible(b, ih, iw, cx, ck, oh, ow, n, kh, kw, stride, pad, in_up,
                                                           int(pixel_shuffle), wstride):
            key = "conv_fprop_upconv"
        if _CLOCK_SHAPES:
            key += f"|B{b} {ih}x{iw}->{oh}x{ow} {c_real}->{n} {kh}x{kw} s{stride} up{in_up}" \
                   f"{' ps' if pixel_shuffle else ''}{' per-sample' if per_sample else ''}|"
    with _lib.on_device(dev), _lib.kernel_clock.span((key, 'bf16' if x.dtype == torch.bfloat16 else 'f32'), flops):
        if residual is not None:
            assert bias is None and act is None and in_up == 1 and not pixel_shuffle
            rv, res_ld = _nhwc_view(residual[0])
            assert rv.shape == (b, n, oh, ow) and rv.dtype == x.dtype
            code = _lib.lib().msg_conv2d_fprop_residual(
                xv.data_ptr(), wk.data_ptr(), y.data_ptr(), _contraction_code(x, mode), b, ih, iw, cx, ck, oh, ow, n, ldy, kh, kw,
                stride, pad, wstride, rv.data_ptr(), res_ld, float(residual[1]), _lib.stream_of(dev))
        elif act is None:
            code = _lib.lib().msg_conv2d_fprop(
                xv.data_ptr(), wk.data_ptr(), _lib.ptr(bias), y.data_ptr(), _contraction_code(x, mode), b, ih, iw, cx, ck, oh,
                ow, n, ldy, kh, kw, stride, pad, in_up, int(pixel_shuffle), wstride, _lib.stream_of(dev))
        else:
            assert bias is None and in_up == 1 and not pixel_shuffle
            act_bias, noise, noise_w, alpha, scale = act[:5]
            _lib.require_gpu(x, act_bias, noise, noise_w)
            mask = None
            if len(act) > 5 and act[5] is not None and kh == 3 and stride == 1 and pad == 1:
                # (act[5]: a list that receives the sign bytes of the output -- (bytes, tile_m, tile_n), the kernel's output
                #  tile -- when the kernel this problem goes to writes them)
                pkey = (x.dtype, b, ih, iw, cx, ck, oh, ow, n, kh, kw, wstride)
                mplan = _PLAN_CACHE.get(pkey)
                if mplan is None:          # (the library's kernel choice is a pure function of the geometry: asked once)
                    mplan = _PLAN_CACHE[pkey] = _lib.lib().msg_conv2d_fprop_plan(_lib.dtype_code(x), b, ih, iw, cx, ck, oh, ow,
                                                                                n, kh, kw, wstride)
                if mplan in (3, 4):
                    from .op_static.fused_act import sign_mask_for
                    mask = sign_mask_for(b, n, oh, ow, x.dtype, dev)
                    if mask is not None:
                        act[5].append((mask, 256 if mplan == 3 else 128, 256 if mplan == 3 else 128))
            code = _lib.lib().msg_conv2d_fprop_act_mask(
                xv.data_ptr(), wk.data_ptr(), y.data_ptr(), _contraction_code(x, mode), b, ih, iw, cx, ck, oh, ow, n, ldy, kh, kw,
                stride, pad, wstride, _lib.ptr(act_bias), _lib.ptr(noise), _lib.ptr(noise_w),
                1 if noise is None else noise.shape[0], float(alpha), float(scale), _lib.ptr(mask), _lib.stream_of(dev))
    _lib.check(code, "msg_conv2d_fprop")
    return y


def _grad_dest(param):
    """The flat-store slice a backward kernel may write ``param``'s gradient into (dist.grad_destination), or None."""
    if param is None or "_msg_grad_slot" not in getattr(param, "__dict__", ()):
        return None
    from .dist import grad_destination
    return grad_destination(param)


def _launch_wgrad(gy, x, o, i, kh, kw, stride, pad, pixel_shuffle, per_sample, low_hw, raw=False, gain=1.0, out=None,
                  mode=None):
    """out: a contiguous fp32 tensor of o*i*kh*kw elements (the parameter's own layout) that receives a SHARED gradient."""
    dev = _lib.require_gpu(gy, x)
    gv, ldgy = _nhwc_view(gy)
    xv, cx = _nhwc_view(x)
    b, _, ih, iw = xv.shape
    oh, ow = low_hw if pixel_shuffle else gv.shape[2:]
    taps = kh * kw
    ldgw = _round_up(i, 4)
    kp = 64 if x.dtype == torch.bfloat16 else 32
    if per_sample:
        # one K sweep per (sample, tile, tap) unless that leaves most of the chip idle (the 512 -> 3 toRGB layers: 64
        # workgroups); then the pixels are split into K-slices
        tiles = ((o + 127) // 128) * ((i + 127) // 128) * taps * b
        k_chunks = max(1, min((oh * ow) // (16 * kp), 1024 // tiles)) if tiles < 256 else 1
    else:
        # (shared weights: the library folds the batch into K and picks the slice count; k_chunks only matters where it cannot)
        tiles = ((o + 127) // 128) * ((i + 127) // 128) * taps * b
        k_chunks = max(1, min((oh * ow + 4 * kp - 1) // (4 * kp), (1024 + tiles - 1) // tiles))
        while b * k_chunks > 65535:
            k_chunks -= 1
    geom = (_contraction_code(x, mode), b, ih, iw, cx, i, oh, ow, ldgy, o, ldgw, kh, kw, stride, pad, int(pixel_shuffle),
            int(per_sample), k_chunks)
    # K-slices that add up to one result meet in a workspace of per-slice slabs and a fixed-order sum (deterministic; no
    # float atomics, no zero fill).  That sum also transposes a SHARED gradient into the parameter's own [O, I, kh, kw]
    # layout, so what autograd accumulates into the flat gradient bucket is a contiguous tensor; per-sample gradients and
    # unsplit results stay in the kernel's [O][tap][I] layout (128-byte runs; the parameter layout would scatter the
    # contraction kernel's stores: measured 25 % slower end to end) and the caller gets a strided view in parameter order.
    need = _WGRAD_WS_CACHE.get(geom)                      # (a pure function of the geometry: one library call per distinct problem)
    if need is None:
        need = _lib.lib().msg_conv2d_wgrad_workspace(*geom)
        if need < 0:
            _lib.check(int(need), "msg_conv2d_wgrad_workspace")
        _WGRAD_WS_CACHE[geom] = need
    ws = _lib.scratch_ptr(need, dev) if need else None        # (launch-scoped: slabs -> the fixed-order sum of the same call)
    if per_sample or raw:
        out = None
    oi_major = (bool(need) or out is not None) and not per_sample and not raw
    if oi_major:
        gw = out.view(o, i, kh, kw) if out is not None else torch.empty((o, i, kh, kw), dtype=torch.float32, device=dev)
    elif per_sample:
        gw = torch.empty((b, o, taps, ldgw), dtype=torch.float32, device=dev)
    else:
        gw = torch.empty((o, taps, ldgw), dtype=torch.float32, device=dev)
    flops = 2.0 * b * oh * ow * o * i * taps
    key = "conv_wgrad"
    if _CLOCK_SHAPES and _lib.kernel_clock.enabled:
        key += f"|B{b} {ih}x{iw}->{oh}x{ow} {i}->{o} {kh}x{kw} s{stride}{' ps' if pixel_shuffle else ''}" \
               f"{' per-sample' if per_sample else ' shared'}{f' slabs{need // (o * taps * ldgw)}' if need else ''}|"
    with _lib.on_device(dev), _lib.kernel_clock.span((key, 'bf16' if x.dtype == torch.bfloat16 else 'f32'), flops):
        code = _lib.lib().msg_conv2d_wgrad(
            gv.data_ptr(), xv.data_ptr(), gw.data_ptr(), *geom, int(oi_major), float(gain), ws, need,
            _lib.stream_of(dev))
    _lib.check(code, "msg_conv2d_wgrad")
    if raw:
        return gw, ldgw                                   # kernel layout [(B)][O][taps][ldgw]
    if oi_major:
        return gw
    lead = gw.shape[:-2]                                  # [(B), O]
    return gw.view(*lead, kh, kw, ldgw)[..., :i].permute(*range(len(lead)), len(lead) + 2, len(lead), len(lead) + 1)


# ------------------------------------------------------------------------------------- the three primitives, raw
class Geometry:
    """kind 'conv': y = conv(x, w, stride, pad);  kind 'up2': y = conv_transpose(x, w^T, kernel 2, stride 2)."""
    __slots__ = ("kind", "kh", "kw", "stride", "pad", "x_hw", "y_hw", "per_sample", "wscale", "mode")

    def __init__(self, kind, kh, kw, stride, pad, x_hw, per_sample, wscale=1.0, mode=None):
        self.kind, self.kh, self.kw, self.stride, self.pad = kind, kh, kw, stride, pad
        self.x_hw, self.per_sample, self.wscale = tuple(x_hw), per_sample, float(wscale)
        # how fp32-storage contractions multiply, fixed when the layer's forward builds its geometry: every launch of the
        # layer's autograd family (F / D / G, first and second order) reads it from here, not from the process-wide switch
        self.mode = FP32_CONTRACTION if mode is None else mode
        if kind == "up2":
            self.y_hw = (2 * x_hw[0], 2 * x_hw[1])
        else:
            self.y_hw = ((x_hw[0] + 2 * pad - kh) // stride + 1, (x_hw[1] + 2 * pad - kw) // stride + 1)


def _oi(w):
    """(out channels, in channels) of a weight in [O,I], [O,I,kh,kw] or [B,O,I,kh,kw] form."""
    return (w.shape[0], w.shape[1]) if w.ndim == 2 else (w.shape[-4], w.shape[-3])


def _relay_fwd_kind(w, dtype, kind):
    wk, ck = _relay_fwd(w, dtype)
    if kind == "up2":                                      # rows n = (2dy+dx)*O + o  <-  w[o, :, dy, dx]
        o = _oi(w)[0]
        wk = wk.transpose(-3, -2).reshape(*wk.shape[:-3], 4 * o, 1, ck).contiguous()
    return wk, ck


def _thin_ok(dtype, i, g: Geometry) -> bool:
    """A 'same' kh x kw conv whose (tap, channel) pairs fit ONE 128-byte K run (the discriminator's 6-channel first layer):
    it runs as a 1x1 conv over the tap-gathered input (msg_gather_taps)."""
    taps = g.kh * g.kw
    return g.kind == "conv" and g.stride == 1 and not g.per_sample and taps > 1 and \
        2 * g.pad + 1 == g.kh and g.kh == g.kw and i * taps <= 128 // (2 if dtype == torch.bfloat16 else 4)


def _gather_taps(x, i, g: Geometry):
    dev = _lib.require_gpu(x)
    xv, cx = _nhwc_view(x)
    b, _, h, w = xv.shape
    ko = 128 // x.element_size()
    out = torch.empty((b, h, w, ko), dtype=x.dtype, device=dev)
    with _lib.on_device(dev):
        code = _lib.lib().msg_gather_taps(xv.data_ptr(), out.data_ptr(), _lib.dtype_code(x), b, h, w, cx, i, g.kh, g.kw,
                                          g.pad, ko, _lib.stream_of(dev))
    _lib.check(code, "msg_gather_taps")
    return out.permute(0, 3, 1, 2), ko


def _relay_thin(w, dtype):
    """[O, I, kh, kw] -> [O, 1, Ko] with K index (tap * I + channel), zero-padded to one 128-byte run."""
    o, i, kh, kw = w.shape
    ko = 128 // (2 if dtype == torch.bfloat16 else 4)
    out = torch.zeros((o, 1, ko), dtype=dtype, device=w.device)
    out[:, 0, :kh * kw * i] = w.permute(0, 2, 3, 1).reshape(o, kh * kw * i)
    return out, ko


def _f_raw(x, w, bias, g: Geometry, act=None, residual=None, out=None):
    if w.ndim == 4 and _thin_ok(x.dtype, w.shape[1], g):
        xc, ko = _gather_taps(x, w.shape[1], g)
        wk, _ = _cached(w, "thin", x.dtype, g.wscale, lambda: _relay_thin(w, x.dtype))
        return _launch_fprop(xc, wk, ko, bias, w.shape[0], g.y_hw, 1, 1, 1, 0, 1, False, False,
                             w.shape[1] * g.kh * g.kw, act=act, residual=residual, out=out, mode=g.mode)
    if g.kind == "up2" and _oi(w)[0] % _vec(x.dtype):
        # the pixel-shuffling epilogue stores whole 16-byte channel vectors per output pixel: pad the output channels
        # with zero filters and drop them again (rare: every up-conv of the models has 512 output channels)
        o_real = _oi(w)[0]
        wp = torch.zeros((*w.shape[:-4], _round_up(o_real, _vec(x.dtype)), *w.shape[-3:]), dtype=w.dtype, device=w.device)
        wp[..., :o_real, :, :, :] = w.detach()
        assert out is None
        return _f_raw(x, wp, bias, g, act=act, residual=residual)[:, :o_real]
    img = _param_images(w, x.dtype, g.wscale, g.kind) if not g.per_sample else None
    wk, ck = img["f"] if img is not None else \
        _cached(w, "f" + g.kind, x.dtype, g.wscale, lambda: _relay_fwd_kind(w, x.dtype, g.kind))
    o, _ = _oi(w)
    if g.kind == "up2":
        assert out is None
        return _launch_fprop(x, wk, ck, None, 4 * o, g.x_hw, 1, 1, 1, 0, 1, True, g.per_sample, _oi(w)[1], mode=g.mode)
    return _launch_fprop(x, wk, ck, bias, o, g.y_hw, g.kh, g.kw, g.stride, g.pad, 1, False, g.per_sample, _oi(w)[1],
                         act=act, residual=residual, out=out, mode=g.mode)


def _act_operands(bias, noise, noise_w, y_shape):
    """fp32, contiguous operands of the fused activation stage (noise [B or 1, 1, H, W] -> [B or 1, H*W])."""
    b32 = None if bias is None else bias.detach().to(torch.float32).contiguous()
    nz = nw = None
    if noise is not None:
        if noise.shape[0] not in (1, y_shape[0]) or noise.shape[1] != 1 or tuple(noise.shape[2:]) != tuple(y_shape[2:]):
            raise _lib.MsgHipError(f"noise shape {tuple(noise.shape)} does not match output {tuple(y_shape)}")
        nz = noise.detach().to(torch.float32).contiguous()
        nw = noise_w.detach().to(torch.float32).contiguous()
    return b32, nz, nw


ACT_BACKWARD_IN_DGRAD = bool(int(os.environ.get("MSG_ACT_BACKWARD_IN_DGRAD", "1")))   # 0: off (A/B; tests compare the two forms)


class ActHandle:
    """Hand-over between a layer that ends in a fused bias (+ noise) + leaky-ReLU stage (the PRODUCER of a map) and the 3x3
    'same' conv that is the map's ONLY consumer, for first-order backward passes: the consumer's data-gradient launch applies
    the stage's backward in its epilogue (msg_conv2d_fprop_act_backward) -- the gradient map between the two backward nodes
    is never written, the stage's bias / noise-weight gradients come out of the same launch -- and leaves them in `done`;
    the producer's backward then takes its incoming gradient as already masked.  The producer fills the fields in forward
    (`arm`); a consumer only hands over to an armed handle; anything the kernel declines keeps the two-pass form."""
    __slots__ = ("armed", "sign", "alpha", "scale", "bias_param", "has_bias", "noise", "done")

    def __init__(self):
        self.armed, self.sign, self.done = False, None, None
        self.alpha = self.scale = 0.0
        self.bias_param, self.has_bias, self.noise = None, False, None

    def arm(self, y, mask, alpha, scale, bias_param, has_bias, noise):
        """y: the stage's stored output; mask: (bytes, tile_m, tile_n) when the forward launch left sign bytes, else None (the
        consumer then reads the signs from its own saved input, which IS y -- the handle must not hold y: the producer's
        node -> handle -> y -> its grad_fn = the producer's node is a cycle only the garbage collector frees, with the whole
        graph's activations hanging off it)."""
        if not ACT_BACKWARD_IN_DGRAD or y.dtype != torch.bfloat16 or not y.is_cuda:
            return
        self.sign = ("mask",) + tuple(mask) if mask is not None else ("map",)
        self.alpha, self.scale = float(alpha), float(scale)
        self.bias_param, self.has_bias, self.noise = bias_param, bool(has_bias), noise
        self.armed, self.done = True, None


_ACTBWD_WS_CACHE: dict = {}


def _launch_dgrad_act_backward(gy, wk, ck, n, kh, kw, per_sample, c_real, handle: ActHandle, residual=None, mode=None,
                               sign_map=None):
    """The data-gradient contraction (a 3x3 'same' conv of gy with the data-gradient weight image) with the backward of the
    activation stage described by `handle` in its epilogue.  Returns the masked gradient and fills handle.done = (grad_bias,
    grad_noise_weight), or returns None when the library declines (another kernel would run this problem, odd layouts)."""
    if torch.is_grad_enabled() or gy.dtype != torch.bfloat16 or kh != 3 or kw != 3 or n % 8:
        return None
    if mode is not None and _contraction_code(gy, mode) != _lib.MSG_BF16:
        return None
    dev = gy.device
    xv, cx = _nhwc_view(gy)
    b, _, h, w_ = xv.shape
    wstride = wk.stride(0) if per_sample else 0
    noise = handle.noise
    wkey = (b, h, w_, cx, ck, n, wstride, noise is not None)
    need = _ACTBWD_WS_CACHE.get(wkey)
    if need is None:
        need = _ACTBWD_WS_CACHE[wkey] = _lib.lib().msg_conv2d_fprop_act_backward_workspace(
            _lib.MSG_BF16, b, h, w_, cx, ck, h, w_, n, kh, kw, wstride, int(noise is not None))
    if not need:
        return None
    sign = handle.sign
    smask = smap = None
    tm = tn = sld = 0
    if sign[0] == "mask":
        smask, tm, tn = sign[1], int(sign[2]), int(sign[3])
        if smask.numel() * 8 != b * n * h * w_ or (tm != 1 and tm % 64):
            return None
    else:
        if sign_map is None or sign_map.dtype != torch.bfloat16:
            return None
        smap, sld = _nhwc_view(sign_map)
        if smap is not sign_map or tuple(smap.shape) != (b, n, h, w_):
            return None
    rv, res_ld = (None, 0)
    if residual is not None:
        assert residual[1] == 1.0
        rv, res_ld = _nhwc_view(residual[0])
        if rv.shape != (b, n, h, w_) or rv.dtype != gy.dtype:
            return None
    y, ldy = _alloc_out(b, n, h, w_, gy.dtype, dev)
    gb = None
    if handle.has_bias:
        bp = handle.bias_param
        if bp is not None and bp.dtype == torch.float32 and bp.shape == (n,):
            gb = _grad_dest(bp)
        if gb is None:
            gb = torch.empty(n, dtype=torch.float32, device=dev)
    nz = gnw = None
    nb = 1
    if noise is not None:
        if noise.shape[0] not in (1, b) or tuple(noise.shape[1:]) != (1, h, w_):
            return None
        nz, nb = noise.detach().to(torch.float32).contiguous(), noise.shape[0]
        gnw = torch.empty(1, dtype=torch.float32, device=dev)
    ws = _lib.scratch_ptr(need, dev)
    flops = 2.0 * b * h * w_ * n * kh * kw * c_real
    key = "conv_fprop_row3_actbwd"
    if _lib.kernel_clock.enabled:                       # (the 256 x 256 tile -- the benchmark's roofline kernel -- or the 128 x 128 one)
        if _lib.lib().msg_conv2d_fprop_plan(_lib.MSG_BF16, b, h, w_, cx, ck, h, w_, n, kh, kw, wstride) == 4:
            key = "conv_fprop_row3n_actbwd"
    if _lib.kernel_clock.enabled and _CLOCK_SHAPES:
        key += f"|B{b} {h}x{w_}->{h}x{w_} {c_real}->{n} 3x3 s1 up1{' per-sample' if per_sample else ''}|"
    with _lib.on_device(dev), _lib.kernel_clock.span((key, 'bf16'), flops):
        code = _lib.lib().msg_conv2d_fprop_act_backward(
            xv.data_ptr(), wk.data_ptr(), y.data_ptr(), _lib.MSG_BF16, b, h, w_, cx, ck, h, w_, n, ldy, kh, kw, 1, 1, wstride,
            _lib.ptr(rv), res_ld, _lib.ptr(smask), tm, tn, _lib.ptr(smap), sld, handle.alpha, handle.scale,
            _lib.ptr(gb), _lib.ptr(nz), nb, _lib.ptr(gnw), ws, need, _lib.stream_of(dev))
    if code == -2:
        _ACTBWD_WS_CACHE[wkey] = 0          # (not asked again for this geometry)
        return None
    _lib.check(code, "msg_conv2d_fprop_act_backward")
    handle.done = (gb, gnw)
    return y


def _d_raw(gy, w, g: Geometry, residual=None, act_bwd: Optional["ActHandle"] = None, sign_map=None):
    """act_bwd: the ActHandle of the layer that produced the conv's input, sign_map: that input (= the layer's stored output)."""
    if act_bwd is not None and act_bwd.armed and not g.per_sample and g.kind == "conv" and g.stride == 1 and g.kh == 3 and \
            g.kw == 3 and g.pad == 1 and w.ndim == 4:
        i = _oi(w)[1]
        img = _param_images(w, gy.dtype, g.wscale, g.kind)
        wk, ok = img["d"] if img is not None else \
            _cached(w, "d" + g.kind, gy.dtype, g.wscale, lambda: _relay_dgrad(w, gy.dtype, flip=True))
        out = _launch_dgrad_act_backward(gy, wk, ok, i, 3, 3, False, _oi(w)[0], act_bwd, residual=residual, mode=g.mode,
                                         sign_map=sign_map)
        if out is not None:
            return out
    return _d_raw_plain(gy, w, g, residual)


def _d_raw_plain(gy, w, g: Geometry, residual=None):
    """Data gradient; residual = (map shaped like the result, gain): (dgrad + map) * gain in the epilogue (plain
    stride-1 convs only -- the caller checks)."""
    assert residual is None or (g.kind == "conv" and g.stride == 1)
    if g.kind == "conv" and g.stride == 2 and _oi(w)[1] % _vec(gy.dtype) == 0:
        return _d_raw_s2(gy, w, g)          # (the pixel-shuffling epilogue needs whole 16-byte channel vectors)
    i = _oi(w)[1]
    if w.ndim == 4 and _thin_ok(gy.dtype, i, g) and i <= 8 and gy.shape[1] >= 64:
        return _d_raw_thin(gy, w, g, residual)
    img = _param_images(w, gy.dtype, g.wscale, g.kind) if not g.per_sample else None
    wk, ok = img["d"] if img is not None else \
        _cached(w, "d" + g.kind, gy.dtype, g.wscale, lambda: _relay_dgrad(w, gy.dtype, flip=g.kind != "up2"))
    if g.kind == "up2":
        return _launch_fprop(gy, wk, ok, None, i, g.x_hw, 2, 2, 2, 0, 1, False, g.per_sample, _oi(w)[0], mode=g.mode)
    pad = g.kh - 1 - g.pad
    assert g.kh == g.kw
    return _launch_fprop(gy, wk, ok, None, i, g.x_hw, g.kh, g.kw, 1, pad, g.stride, False, g.per_sample, _oi(w)[0],
                         residual=residual, mode=g.mode)


def _relay_thin_dgrad(w, dtype):
    """[O, I, kh, kw] -> [Ko, 1, Ok]: row k = tap * I + c holds w[:, c, tap] -- the weights of the 1x1 contraction from the
    output gradient to the tap-gathered input's gradient (the transpose of _relay_thin)."""
    o, i, kh, kw = w.shape
    esz = 2 if dtype == torch.bfloat16 else 4
    ko, ok = 128 // esz, _round_up(o, 128 // esz)
    out = torch.zeros((ko, 1, ok), dtype=dtype, device=w.device)
    out[:kh * kw * i, 0, :o] = w.permute(2, 3, 1, 0).reshape(kh * kw * i, o)
    return out, ok


def _d_raw_thin(gy, w, g: Geometry, residual=None):
    """Data gradient of a few-channel 'same' conv (see _thin_ok) as the adjoint of its tap-gathered forward: ONE 1x1
    contraction O -> Ko over gy, then msg_fold_taps brings the Ko = taps x I planes back to the I channels (adding the
    residual map on the way).  Replaces nine K-steps per tile on MFMA tiles that are 95 % padding (3x3 128 -> 6 @256^2, batch
    16: 392 us, 37 TFLOP/s) by two streaming passes."""
    o, i = w.shape[0], w.shape[1]
    dev = gy.device
    wd, ok = _cached(w, "dthin", gy.dtype, g.wscale, lambda: _relay_thin_dgrad(w, gy.dtype))
    ko = 128 // gy.element_size()
    flops = 2.0 * gy.shape[0] * g.x_hw[0] * g.x_hw[1] * o * i * g.kh * g.kw
    gk = _launch_fprop(gy, wd, ok, None, ko, g.x_hw, 1, 1, 1, 0, 1, False, False, o, flops=flops, mode=g.mode)
    gkv, ldk = _nhwc_view(gk)
    b, _, h, w_ = gkv.shape
    gx, ldx = _alloc_out(b, i, h, w_, gy.dtype, dev)
    add = ld_add = None
    if residual is not None:
        assert residual[1] == 1.0
        add, ld_add = _nhwc_view(residual[0])
        assert add.shape == gx.shape and add.dtype == gx.dtype
    with _lib.on_device(dev), _lib.kernel_clock.span(("fold_taps", gy.dtype), (gkv.numel() + b * h * w_ * ldx) * gy.element_size()):
        code = _lib.lib().msg_fold_taps(gkv.data_ptr(), _lib.ptr(add), gx.data_ptr(), _lib.dtype_code(gy), b, h, w_, ldk, i,
                                        g.kh, g.kw, g.pad, ldx, ld_add or 0, _lib.stream_of(dev))
    _lib.check(code, "msg_fold_taps")
    return gx


def _d_raw_s2(gy, w, g: Geometry):
    """Data gradient of a stride-2 conv through the parity decomposition (see _relay_dgrad_s2)."""
    o, i = _oi(w)
    wk, ok, taps, pad2 = _cached(w, "ds2", gy.dtype, g.wscale, lambda: _relay_dgrad_s2(w, gy.dtype, g.pad))
    hc, wc = (g.x_hw[0] + 1) // 2, (g.x_hw[1] + 1) // 2
    flops = 2.0 * gy.shape[0] * gy.shape[2] * gy.shape[3] * o * i * g.kh * g.kw
    gx = _launch_fprop(gy, wk, ok, None, 4 * i, (hc, wc), taps, taps, 1, pad2, 1, True, g.per_sample, o, flops=flops, mode=g.mode)
    return gx[:, :, :g.x_hw[0], :g.x_hw[1]]


def _g_raw(gy, x, o, i, g: Geometry, out=None):
    if _thin_ok(x.dtype, i, g):
        # weight gradient of the tap-gathered 1x1 form: gy is read once (not once per tap), one channel tile
        xc, ko = _gather_taps(x, i, g)
        gwp = _launch_wgrad(gy, xc, o, ko, 1, 1, 1, 0, False, False, None, gain=g.wscale, mode=g.mode)      # [O, Ko, 1, 1]
        taps = g.kh * g.kw
        return gwp[:, :taps * i, 0, 0].reshape(o, taps, i).permute(0, 2, 1).reshape(o, i, g.kh, g.kw)
    if g.kind == "up2":
        return _launch_wgrad(gy, x, o, i, 2, 2, 1, 0, True, g.per_sample, g.x_hw, gain=g.wscale, out=out, mode=g.mode)
    return _launch_wgrad(gy, x, o, i, g.kh, g.kw, g.stride, g.pad, False, g.per_sample, None, gain=g.wscale, out=out, mode=g.mode)


def _consumed(ctx, arg_index: int, tensor_ordinal: int) -> bool:
    """Whether the gradient of forward argument `arg_index` (the `tensor_ordinal`-th TENSOR argument) is worth computing
    in this backward call: it requires grad AND the engine is going to run the node it would be handed to.  In a partial
    backward -- ``torch.autograd.grad(outputs, inputs=[images])``, the first pass of the R1 and path-length regularisers --
    ``ctx.needs_input_grad`` still says True for every weight, although their gradients are thrown away on return; asking
    the engine (the query activation checkpointing uses for its early stop) skips the weight-gradient contractions of the
    whole discriminator there.  Any doubt (no graph task, a leaf that is itself one of the requested inputs) -> True."""
    if not ctx.needs_input_grad[arg_index]:
        return False
    try:
        node = ctx.next_functions[tensor_ordinal][0]
        return node is None or bool(torch._C._will_engine_execute_node(node))
    except Exception:
        return True


# ------------------------------------------------------------------------------------- autograd closure of F/D/G
class _ConvF(Function):
    @staticmethod
    def forward(ctx, x, w, bias, g):
        ctx.g = g
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w)
        return _f_raw(x, w, bias, g)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g = ctx.g
        gx = _derive(_ConvD, gy, w, g) if ctx.needs_input_grad[0] else None
        gw = _derive(_ConvG, gy, x, _oi(w), w.ndim, g, w) if _consumed(ctx, 1, 1) else None
        gb = _channel_sums(gy) if ctx.has_bias and _consumed(ctx, 2, 2) else None
        return gx, gw, gb, None


def _channel_sums(gy):
    """gy.sum((0, 2, 3)) in fp32: on the activation backward's partial-sum kernels for dense channels-last maps (fixed
    order, a third of the library reduction's time on the strided convs' maps), the library otherwise / in second-order graphs."""
    c = gy.shape[1]
    vec = _vec(gy.dtype)
    if torch.is_grad_enabled() or not gy.is_cuda or gy.dtype not in (torch.float32, torch.bfloat16) or c % vec or \
            not gy.is_contiguous(memory_format=torch.channels_last) or gy.data_ptr() % 16 or gy.numel() == 0:
        # (the cast inside the reduction: `gy.float()` would materialise an fp32 copy of the whole map first)
        return gy.sum(dim=(0, 2, 3), dtype=torch.float32)
    dev = gy.device
    need = _lib.lib().msg_bias_act_backward_workspace(gy.numel(), 1, c, 0)
    ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
    out = torch.empty(c, dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        code = _lib.lib().msg_channel_sums(gy.data_ptr(), out.data_ptr(), _lib.dtype_code(gy), gy.numel(), c, ws.data_ptr(),
                                           need, _lib.stream_of(dev))
    _lib.check(code, "msg_channel_sums")
    return out


def _dgrad_add_ok(add, x_shape, dtype, g) -> bool:
    """Whether `add` (a map shaped like the data gradient) can ride in the data-gradient launch's residual epilogue."""
    return add is not None and g.kind == "conv" and g.stride == 1 and tuple(add.shape) == tuple(x_shape) and add.dtype == dtype


class _ConvD(Function):
    """gx = D(gy, w) [+ add].  `add` (plain stride-1 convs): a map that is added in the launch's residual epilogue -- the
    differentiable form of the hand-overs that first-order backward does with raw launches, for the graphs R1 differentiates
    again: where a block input has two consumers, the second data gradient accumulates INTO the first instead of autograd (or
    a node of this package) adding two full maps in a pass of its own (round 5; before, every such sum of a create_graph
    backward was a stock elementwise launch: 3.6 GB per regularised iteration at 256^2)."""

    @staticmethod
    def forward(ctx, gy, w, g, add=None):
        ctx.g = g
        ctx.save_for_backward(gy, w)
        return _d_raw(gy, w, g, residual=None if add is None else (add, 1.0))

    @staticmethod
    def backward(ctx, v):
        gy, w = ctx.saved_tensors
        g = ctx.g
        ggy = _derive(_ConvF, v, w, None, g) if ctx.needs_input_grad[0] else None
        gw = _derive(_ConvG, gy, v, _oi(w), w.ndim, g) if ctx.needs_input_grad[1] else None
        return ggy, gw, None, (v if len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3] else None)


RESIDUAL_FORK_NODE = True        # False: two consumers of the merged gradient, autograd adds their cotangents (tests compare)


class _ConvDFork(Function):
    """(D(gy, w), gy): the residual merge's backward in a graph that is differentiated again -- gy goes on to the main branch
    unchanged AND through the 1x1 residual conv's data gradient.  As two consumers of one tensor their cotangents met in a
    stock add over the block's output map (R1, once per discriminator block); as one node the cotangent of gy is the
    forward conv of the first with the second added in the launch's residual epilogue."""

    @staticmethod
    def forward(ctx, gy, w, g):
        ctx.g = g
        ctx.save_for_backward(gy, w)
        return _d_raw(gy, w, g), gy.view_as(gy)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v, c):
        gy, w = ctx.saved_tensors
        g = ctx.g
        ggy = gw = None
        if v is None:
            return c, None, None
        if ctx.needs_input_grad[0]:
            if c is not None and g.kind == "conv" and g.stride == 1 and c.shape == gy.shape and c.dtype == v.dtype:
                ggy = _f_raw(v, w, None, g, residual=(c, 1.0))
            else:
                ggy = _f_raw(v, w, None, g)
                ggy = ggy if c is None else ggy + c
        if ctx.needs_input_grad[1]:
            gw = _derive(_ConvG, gy, v, _oi(w), w.ndim, g)
        return ggy, gw, None


class _ConvG(Function):
    @staticmethod
    def forward(ctx, gy, x, oi, w_ndim, g, dest=None):
        """dest: the PARAMETER this is the gradient of -- in a first-order backward its slice of the flat gradient store is
        written directly when that is possible (conv_ops._grad_dest)."""
        ctx.g = g
        ctx.save_for_backward(gy, x)
        out = None if (g.per_sample or _thin_ok(x.dtype, oi[1], g)) else _grad_dest(dest)
        gw = _g_raw(gy, x, oi[0], oi[1], g, out=out)
        if out is not None:
            return out                        # (already shaped like the parameter)
        if w_ndim == 5 and not g.per_sample:  # a shared weight stored [1, O, I, kh, kw] (the generator's modulated convs)
            return gw.unsqueeze(0)
        return gw.reshape(oi) if w_ndim == 2 else gw

    @staticmethod
    def backward(ctx, u):
        gy, x = ctx.saved_tensors
        g = ctx.g
        ggy = _derive(_ConvF, x, u, None, g) if ctx.needs_input_grad[0] else None
        gx = _derive(_ConvD, gy, u, g) if ctx.needs_input_grad[1] else None
        return ggy, gx, None, None, None, None


class _ConvActF(Function):
    """conv -> (noise +) bias -> leaky ReLU in ONE launch (activation in the conv epilogue).  The backward is composed
    of the differentiable pieces that the two-pass form uses (activation backward from the OUTPUT's sign, then the
    D / G contractions), so first- and second-order gradients are those of conv followed by FusedLeakyReLU."""

    @staticmethod
    def forward(ctx, x, w, act_bias, noise, noise_w, g, alpha, scale, slot=None, out_scale=None, act_handle=None,
                input_act=None):
        o = _oi(w)[0]
        b32, nz, nw = _act_operands(act_bias, noise, noise_w, (x.shape[0], o, *g.y_hw))
        holder = [] if any(ctx.needs_input_grad) else None   # (a forward that no backward follows writes no sign bytes)
        y = _f_raw(x, w, None, g, act=(b32, nz, nw, alpha, scale, holder))
        ctx.mask = holder[0] if holder else None            # sign bytes of y, when the forward kernel wrote them
        ctx.slot, ctx.out_scale = slot, out_scale
        # act_handle: this layer's output has ONE consumer, a 3x3 conv that may apply this activation's backward in its
        # data-gradient epilogue (ActHandle); input_act: the handle of the layer that produced x
        ctx.act_handle = ctx.input_act = None
        if act_handle is not None and out_scale is None and noise is None and holder is not None:
            act_handle.arm(y, ctx.mask, alpha, scale, act_bias, act_bias is not None, None)
            ctx.act_handle = act_handle if act_handle.armed else None
        if input_act is not None and input_act.armed:
            ctx.input_act = input_act
        ctx.bias_param = act_bias
        ctx.g, ctx.cfg = g, (alpha, scale, act_bias is not None, noise is not None)
        ctx.nw_shape = None if noise_w is None else noise_w.shape
        ctx.save_for_backward(x, w, y, noise)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .op_static.fused_act import FusedLeakyReLUFunctionBackward
        x, w, y, noise = ctx.saved_tensors
        alpha, scale, has_bias, has_noise = ctx.cfg
        g = ctx.g
        owed = 1.0
        if ctx.out_scale is not None and ctx.out_scale.pending is not None:
            owed, ctx.out_scale.pending = ctx.out_scale.pending, None        # (see GradScale: the consumer's gain, deferred)
        handed = ctx.act_handle.done if ctx.act_handle is not None else None
        if handed is not None:
            # the consumer's data-gradient launch applied this activation's backward in its epilogue (ActHandle): gy IS gpre
            ctx.act_handle.done = None
            gpre, (gb, gnw) = gy, handed
        else:
            gpre, gb, gnw = _derive(FusedLeakyReLUFunctionBackward, gy, y, noise if has_noise else None,
                                    ctx.bias_param if has_bias else False, alpha, scale * owed, ctx.mask)
        gx = None
        if ctx.needs_input_grad[0]:
            other = ctx.slot.g if ctx.slot is not None else None
            pre = ctx.input_act if not torch.is_grad_enabled() else None      # (x's producer: its activation backward rides along)
            if _dgrad_add_ok(other, x.shape, gpre.dtype, g):
                # the block input's OTHER gradient (from the 1x1 residual conv, computed just before) is added in this
                # data-gradient conv's epilogue: no separate accumulation pass over the input map (second-order graphs: the
                # same launch as a differentiable node)
                gx = _derive(_ConvD, gpre, w, g, other) if torch.is_grad_enabled() else \
                    _d_raw(gpre, w, g, residual=(other, 1.0), act_bwd=pre, sign_map=x)
                ctx.slot.merged = True
            elif pre is not None:
                gx = _d_raw(gpre, w, g, act_bwd=pre, sign_map=x)
            else:
                gx = _derive(_ConvD, gpre, w, g)
        gw = _derive(_ConvG, gpre, x, _oi(w), w.ndim, g, w) if _consumed(ctx, 1, 1) else None
        return gx, gw, (gb if has_bias and ctx.needs_input_grad[2] else None), None, \
            (gnw.reshape(ctx.nw_shape) if has_noise and ctx.needs_input_grad[4] else None), None, None, None, None, None, None, None


class _ConvResidualF(Function):
    """y = (conv(x, w) + main) * gain in one launch (the residual merge of a discriminator block in the epilogue of its
    1x1 residual conv), optionally handed out twice (fork) for an output with two consumers.  Backward: the merged,
    rescaled gradient goes to `main` as it is and through the D / G contractions to x and w."""

    @staticmethod
    def forward(ctx, x, w, main, g, gain, fork, slot=None, main_scale=None, out=None):
        y = _f_raw(x, w, None, g, residual=(main, gain), out=out)
        if out is not None:
            y = out.view_as(out)                   # (a fresh alias: `out` itself is an input of this node)
        ctx.slot, ctx.main_scale = slot, main_scale
        ctx.g, ctx.gain, ctx.fork = g, float(gain), fork
        ctx.save_for_backward(x, w)
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    def backward(ctx, *grads):
        from .op_static.fused_act import _ScaledAddRows, _rows_ok
        x, w = ctx.saved_tensors
        g1 = grads[0]
        g2 = grads[1] if ctx.fork else None
        if (g1 is None or g2 is None) and ctx.main_scale is not None and ctx.needs_input_grad[2]:
            # one incoming gradient: nothing to add, and the rescaling is deferred -- `main`'s activation backward
            # multiplies the gain in (GradScale), the two contractions take it through their weight scale.  Second-order graphs
            # too (round 5): every piece the gain moves into is a differentiable node with the factor as a constant (_ConvD /
            # _ConvG through their geometry's wscale, the activation backward through its scale), and a create_graph backward
            # runs its nodes in the same order, so `pending` meets the same consumer
            gs = g1 if g1 is not None else g2
            ctx.main_scale.pending = ctx.gain
            gg = Geometry(ctx.g.kind, ctx.g.kh, ctx.g.kw, ctx.g.stride, ctx.g.pad, ctx.g.x_hw, ctx.g.per_sample,
                          ctx.g.wscale * ctx.gain, mode=ctx.g.mode)
            gs_main = gs
            if torch.is_grad_enabled() and ctx.needs_input_grad[0] and RESIDUAL_FORK_NODE:
                gx, gs_main = _ConvDFork.apply(gs, w, gg)          # (one node: the two cotangents of gs meet in a conv epilogue)
            else:
                gx = _derive(_ConvD, gs, w, gg) if ctx.needs_input_grad[0] else None
            if ctx.slot is not None and gx is not None:
                ctx.slot.g = gx
            gw = _derive(_ConvG, gs, x, _oi(w), w.ndim, gg, w) if _consumed(ctx, 1, 1) else None
            return gx, gw, gs_main, None, None, None, None, None, None
        if g1 is None or g2 is None:
            gs = (g1 if g1 is not None else g2) * ctx.gain
        elif _rows_ok(g1, g2):
            gs = _derive(_ScaledAddRows, g1, g2, ctx.gain)
        else:
            gs = (g1 + g2) * ctx.gain
        gx = _derive(_ConvD, gs, w, ctx.g) if ctx.needs_input_grad[0] else None
        if ctx.slot is not None and gx is not None:
            ctx.slot.g = gx                  # the main branch's first conv adds it in its data-gradient epilogue
        gw = _derive(_ConvG, gs, x, _oi(w), w.ndim, ctx.g, w) if _consumed(ctx, 1, 1) else None
        return gx, gw, (gs if ctx.needs_input_grad[2] else None), None, None, None, None, None, None


class _MultiConvF(Function):
    """Several bias-free stride-1 convs of ONE input (the non-local block's theta / phi / g / residual projections).  In
    backward the data gradients are chained through the residual epilogue -- each conv's data-gradient launch adds the
    sum so far -- instead of autograd adding n full maps in n - 1 separate passes.  Second-order graphs take the plain
    differentiable form."""

    @staticmethod
    def forward(ctx, x, geoms, *weights):
        ctx.geoms = geoms
        ctx.save_for_backward(x, *weights)
        return tuple(_f_raw(x, w, None, g) for w, g in zip(weights, geoms))

    @staticmethod
    def backward(ctx, *grads):
        x, *weights = ctx.saved_tensors
        second_order = torch.is_grad_enabled()
        gx, gws = None, []
        for k, (gy, w, g) in enumerate(zip(grads, weights, ctx.geoms)):
            if gy is None:
                gws.append(None)
                continue
            if ctx.needs_input_grad[0]:
                if gx is None or not _dgrad_add_ok(gx, x.shape, gy.dtype, g):
                    part = _derive(_ConvD, gy, w, g)
                    gx = part if gx is None else gx + part
                elif second_order:
                    gx = _derive(_ConvD, gy, w, g, gx)              # (differentiable: the sum so far rides in the epilogue)
                else:
                    gx = _d_raw(gy, w, g, residual=(gx, 1.0))
            gws.append(_derive(_ConvG, gy, x, _oi(w), w.ndim, g, w) if _consumed(ctx, 2 + k, 1 + k) else None)
        return (gx, None, *gws)


def conv2d_shared_input(x, weights_and_scales, padding=0):
    """[conv(x, wscale * weight) for (weight, wscale) in weights_and_scales] -- bias-free, stride 1 -- as one autograd
    node whose backward accumulates the input's gradient inside the data-gradient launches (see _MultiConvF)."""
    p = _square(padding, "padding")
    geoms = tuple(Geometry("conv", w.shape[2], w.shape[3], 1, p, x.shape[2:], False, ws) for w, ws in weights_and_scales)
    return _MultiConvF.apply(x, geoms, *[w for w, _ in weights_and_scales])


class GradScale:
    """A gain that the consumer of an activation's output owes its gradient, handed to the activation's own backward
    instead of being applied in a pass of its own: the residual merge y = (conv1x1(x) + main) * gain sends `main` the
    gradient gy * gain; when `main` is the output of a fused conv + leaky ReLU whose ONLY consumer is that merge (the
    discriminator block), the merge's backward passes gy on untouched and leaves `gain` here, and the activation
    backward -- a pass over the same map anyway -- multiplies it in (its `scale` argument).  First- and (round 5) second-order
    graphs alike: the factor only ever enters differentiable nodes as a constant."""
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None


class GradSlot:
    """Hand-over point between the two consumers of a block input (see fork_input)."""
    __slots__ = ("g", "merged")

    def __init__(self):
        self.g, self.merged = None, False


class HeadGradSlot:
    """Hand-over between the two consumers' producer and the image head of a generator level (first-order backward only).
    The styled layer's output feeds the next level and the level's head, a 1x1 modulated conv to <= 8 planes; the head's
    backward runs first (it is a consumer) and, instead of writing its data gradient as a full map for autograd to sum,
    leaves here what that gradient is made of -- its own incoming gradient, base weights, style, scale.  The styled
    layer's activation backward forms it on the fly (op_static.fused_act.act_backward_with_head).  `head` is consumed
    once; `dgrad` computes the ordinary data gradient for the cases the fused kernel declines."""
    __slots__ = ("head", "dgrad", "armed")

    def __init__(self):
        self.head = None
        self.dgrad = None
        self.armed = False          # set by the producer's forward: a head only hands over to a producer that will look

    def take(self):
        head, dgrad, self.head, self.dgrad = self.head, self.dgrad, None, None
        return head, dgrad


class _ForkInput(Function):
    """x -> two aliases for a discriminator block input read by the main branch's first conv AND by the 1x1 residual
    conv.  In backward the residual conv runs first (it is the last node of the block), leaves its input gradient in
    the slot, and the main branch's first conv -- the last node of the block to run -- adds it in the epilogue of its
    data-gradient conv; this node then passes that sum on instead of adding two maps in a separate pass.  Whenever
    the hand-over did not happen (second-order graphs, other shapes, another order) it adds them itself."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g_main, g_res):
        slot = ctx.slot
        merged, slot.g, slot.merged = slot.merged, None, False
        if merged or g_res is None:
            return g_main, None
        if g_main is None:
            return g_res, None
        return g_main + g_res, None


def _square(value, what: str) -> int:
    """The kernels' Geometry carries one stride / padding for both axes; the reference's (h, w) tuples
    (equalized_layer.py:18-21) are accepted when both entries agree and refused -- not silently truncated -- otherwise."""
    if isinstance(value, int):
        return value
    value = tuple(value)
    if len(value) != 2 or value[0] != value[1]:
        raise _lib.MsgHipError(f"conv2d: only square {what} is implemented, got {value}")
    return int(value[0])


def fork_input(x, slot: GradSlot):
    return _ForkInput.apply(x, slot)


def conv2d_add_residual(x, weight, main, gain, stride=1, padding=0, wscale=1.0, fork=False, grad_slot=None,
                        main_grad_scale=None, out=None):
    """(conv(x, wscale * weight) + main) * gain; with fork=True two aliases of the result (see scaled_add_fork).
    ``main_grad_scale``: the GradScale that `main`'s producer was given (see there).  ``out``: a destination from
    cat_destination -- the result is written there and the returned tensor(s) alias it."""
    s, p = _square(stride, "stride"), _square(padding, "padding")
    g = Geometry("conv", weight.shape[2], weight.shape[3], s, p, x.shape[2:], False, wscale)
    return _ConvResidualF.apply(x, weight, main, g, float(gain), bool(fork), grad_slot, main_grad_scale, out)


# ------------------------------------------------------------------------------------------------- public entry
def conv2d_bias_act(x, weight, act_bias, stride=1, padding=0, wscale=1.0, negative_slope=0.2, scale=1.0,
                    grad_slot=None, out_grad_scale=None, act_handle=None, input_act=None):
    """leaky_relu(conv(x, wscale * weight) + act_bias) * scale in one launch (EqualizedConv2d -> FusedLeakyReLU).
    ``out_grad_scale``: a GradScale shared with the output's only consumer (see there).  ``act_handle``: an ActHandle shared
    with the output's only consumer, a 3x3 conv that receives it as ``input_act`` (see ActHandle)."""
    s, p = _square(stride, "stride"), _square(padding, "padding")
    g = Geometry("conv", weight.shape[2], weight.shape[3], s, p, x.shape[2:], False, wscale)
    return _ConvActF.apply(x, weight, act_bias, None, None, g, float(negative_slope), float(scale), grad_slot,
                           out_grad_scale, act_handle, input_act)


def conv2d(x, weight, bias=None, stride=1, padding=0, wscale=1.0):
    """Shared-weight conv: y = conv(x, wscale * weight) + bias; weight [O,I,kh,kw] fp32, x [B,I,H,W]; y in x's dtype.
    Passing the raw parameter plus its equalized-lr scale lets the re-laid weights be cached between optimizer steps."""
    s, p = _square(stride, "stride"), _square(padding, "padding")
    g = Geometry("conv", weight.shape[2], weight.shape[3], s, p, x.shape[2:], False, wscale)
    return _ConvF.apply(x, weight, None if bias is None else bias.float(), g)


def conv_transpose2d_2x2(x, weight, bias=None, wscale=1.0):
    """F.conv_transpose2d(x, wscale * weight, bias, stride=2, padding=0) for a 2x2 kernel, weight [I, O, 2, 2] (torch's
    transposed-conv layout): the non-overlapping case, four sub-pixel 1x1 contractions written pixel-shuffled (the
    'up2' geometry of the generator's up-convs, with shared weights)."""
    if tuple(weight.shape[2:]) != (2, 2):
        raise _lib.MsgHipError("conv_transpose2d_2x2: 2x2 kernel, stride 2, padding 0 only")
    g = Geometry("up2", 2, 2, 1, 1, x.shape[2:], False, wscale)
    y = _ConvF.apply(x, weight.transpose(0, 1), None, g)
    return y if bias is None else y + bias.view(1, -1, 1, 1).to(y.dtype)


_LINEAR_MAX_ROWS = 256        # above this the batch rows are worth an MFMA tile: the conv path takes over


def _lin_call(name, flops, *args):
    dev = args[0].device
    with _lib.on_device(dev), _lib.kernel_clock.span((name, "f32"), flops):
        code = getattr(_lib.lib(), f"msg_{name}")(*[a.data_ptr() if isinstance(a, torch.Tensor) else
                                                   (0 if a is None else a) for a in args], _lib.stream_of(dev))
    _lib.check(code, f"msg_{name}")


def _dense32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _lib.MsgHipError("the few-row linear kernels are fp32 (mapping network / style affines / heads)")
    return [None if t is None else t.contiguous() for t in ts]


class _LinF(Function):
    """y = gain * x @ w^T (+ bias).  csrc/linear.hip; derivatives are _LinD / _LinG (closed family, any order)."""

    @staticmethod
    def forward(ctx, x, w, bias, gain, bias_gain=1.0):
        _lib.require_gpu(x, w, bias)
        x, w, bias = _dense32(x, w, bias)
        (m, k), n = x.shape, w.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        _lin_call("linear_fprop", 2.0 * m * n * k, x, w, bias, y, m, n, k, float(gain), float(bias_gain))
        ctx.save_for_backward(x, w)
        ctx.gain, ctx.has_bias, ctx.bias_gain = float(gain), bias is not None, float(bias_gain)
        ctx.bias_param = bias
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        gx = _derive(_LinD, gy, w, ctx.gain) if need_x else None
        gw = gb = None
        if need_w and need_b and ctx.has_bias and not torch.is_grad_enabled():
            (gy_,) = _dense32(gy)                     # first-order step: weight and bias gradient in one launch
            (m, n), k = gy_.shape, x.shape[1]
            gw, gb = _grad_dest(w), _grad_dest(ctx.bias_param)        # (their slices of the flat gradient store, if free)
            gw = gw if gw is not None else torch.empty((n, k), dtype=torch.float32, device=x.device)
            gb = gb if gb is not None else torch.empty((n,), dtype=torch.float32, device=x.device)
            _lin_call("linear_wgrad", 2.0 * m * n * k, gy_, x, gw, gb, m, n, k, ctx.gain, ctx.bias_gain)
        else:
            if need_w:
                gw = _derive(_LinG, gy, x, ctx.gain)
            if need_b and ctx.has_bias:
                gb = gy.sum(dim=0) if ctx.bias_gain == 1.0 else gy.sum(dim=0) * ctx.bias_gain
        return gx, gw, gb, None, None


class _LinD(Function):
    """gx = gain * gy @ w."""

    @staticmethod
    def forward(ctx, gy, w, gain):
        _lib.require_gpu(gy, w)
        gy, w = _dense32(gy, w)
        (m, n), k = gy.shape, w.shape[1]
        gx = torch.empty((m, k), dtype=torch.float32, device=gy.device)
        _lin_call("linear_dgrad", 2.0 * m * n * k, gy, w, gx, m, n, k, float(gain))
        ctx.save_for_backward(gy, w)
        ctx.gain = float(gain)
        return gx

    @staticmethod
    def backward(ctx, g):
        gy, w = ctx.saved_tensors
        d_gy = _derive(_LinF, g, w, None, ctx.gain) if ctx.needs_input_grad[0] else None
        d_w = _derive(_LinG, gy, g, ctx.gain) if ctx.needs_input_grad[1] else None
        return d_gy, d_w, None


class _LinG(Function):
    """gw = gain * gy^T @ x."""

    @staticmethod
    def forward(ctx, gy, x, gain):
        _lib.require_gpu(gy, x)
        gy, x = _dense32(gy, x)
        (m, n), k = gy.shape, x.shape[1]
        gw = torch.empty((n, k), dtype=torch.float32, device=gy.device)
        _lin_call("linear_wgrad", 2.0 * m * n * k, gy, x, gw, None, m, n, k, float(gain), 1.0)
        ctx.save_for_backward(gy, x)
        ctx.gain = float(gain)
        return gw

    @staticmethod
    def backward(ctx, g):
        gy, x = ctx.saved_tensors
        d_gy = _derive(_LinF, x, g, None, ctx.gain) if ctx.needs_input_grad[0] else None
        d_x = _derive(_LinD, gy, g, ctx.gain) if ctx.needs_input_grad[1] else None
        return d_gy, d_x, None


def linear(x, weight, bias=None, wscale=1.0, bias_scale=1.0):
    """x [B,I] @ (wscale * weight[O,I])^T (+ bias_scale * bias).  Few fp32 rows (the mapping network, the style affines, the
    classification head) go to the one-launch kernels of csrc/linear.hip; anything else is the same contraction as a
    1x1 convolution with the batch rows as 'pixels' of one sample."""
    b, i = x.shape
    o = weight.shape[0]
    if x.dtype == torch.float32 and weight.dtype == torch.float32 and b <= _LINEAR_MAX_ROWS:
        return _LinF.apply(x, weight, None if bias is None else bias.float(), float(wscale), float(bias_scale))
    if bias is not None and bias_scale != 1.0:
        bias = bias * bias_scale
    g = Geometry("conv", 1, 1, 1, 0, (b, 1), False, wscale)
    y = _ConvF.apply(x.reshape(1, b, 1, i).permute(0, 3, 1, 2), weight, None if bias is None else bias.float(), g)
    return y.permute(0, 2, 3, 1).reshape(b, o)


_PTR_TABLES: dict = {}


def _ptr_table(tensors, dev):
    """Device int64 table of the tensors' addresses (cached: parameters keep their storage between steps)."""
    key = (dev.index, tuple(t.data_ptr() for t in tensors))
    hit = _PTR_TABLES.get(key)
    if hit is None:
        if len(_PTR_TABLES) > 64:
            _PTR_TABLES.clear()
        hit = _PTR_TABLES[key] = torch.tensor(key[1], dtype=torch.int64, device=dev)
    return hit


class _GroupedLinear(Function):
    """out[g] = wscale * latent[:, slot[g]] @ W_g^T + bias_scale * b_g for G same-shaped layers in ONE launch (and two
    in backward) -- see msg_linear_grouped_fprop.  Inputs: latent [B,L,K], slot (tuple), then G weights and G biases.
    Higher-order requests differentiate the per-layer composite instead."""

    @staticmethod
    def forward(ctx, latent, slot, wscale, bias_scale, *wb):
        g = len(slot)
        ws, bs = wb[:g], wb[g:]
        dev = _lib.require_gpu(latent, *ws, *bs)
        (lat,) = _dense32(latent)
        b, l, k = lat.shape
        n = ws[0].shape[0]
        for w, bb in zip(ws, bs):
            if w.shape != (n, k) or bb.shape != (n,) or w.dtype != torch.float32 or bb.dtype != torch.float32 \
                    or not w.is_contiguous():
                raise _lib.MsgHipError("grouped linear: layers must be fp32, contiguous and of one shape")
        slot_t = _ptr_table_ints(slot, dev)
        y = torch.empty((g, b, n), dtype=torch.float32, device=dev)
        with _lib.on_device(dev), _lib.kernel_clock.span("linear_grouped_fprop/f32", 2.0 * g * b * n * k):
            code = _lib.lib().msg_linear_grouped_fprop(lat.data_ptr(), slot_t.data_ptr(), _ptr_table(ws, dev).data_ptr(),
                                                       _ptr_table(bs, dev).data_ptr(), y.data_ptr(), g, b, n, k, l,
                                                       float(wscale), float(bias_scale), _lib.stream_of(dev))
        _lib.check(code, "msg_linear_grouped_fprop")
        ctx.save_for_backward(latent, *ws, *bs)
        ctx.cfg = (tuple(slot), float(wscale), float(bias_scale))
        return y

    @staticmethod
    def backward(ctx, gy):
        slot, wscale, bias_scale = ctx.cfg
        g = len(slot)
        lat, ws, bs = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + g], ctx.saved_tensors[1 + g:]
        need = ctx.needs_input_grad
        if torch.is_grad_enabled() and need[0] and not any(_consumed(ctx, 4 + j, 1 + j) for j in range(2 * g) if need[4 + j]):
            # a second-order graph that wants the LATENT's gradient only (the path-length regulariser differentiates
            # d image / d latent: multi_stylegan_generator.py:193-200): one differentiable node on the grouped kernels instead
            # of 20 per-layer linears recomputed and differentiated one by one (round 5: ~230 launches of 13-22 us per
            # regularised iteration on an otherwise idle chip)
            glat = _GroupedLinD.apply(gy, slot, wscale, lat.shape[1], *ws)
            return (glat, None, None, None, *([None] * (2 * g)))
        if torch.is_grad_enabled():
            with torch.enable_grad():
                ins = [t for t, nd in zip((lat, *ws, *bs), (need[0], *need[4:])) if nd]
                out = torch.stack([linear(lat[:, slot[j]], ws[j], bs[j], wscale, bias_scale) for j in range(g)])
                grads = list(torch.autograd.grad(out, ins, gy, create_graph=True, allow_unused=True))
            res = [grads.pop(0) if nd else None for nd in (need[0], *need[4:])]
            return (res[0], None, None, None, *res[1:])
        dev = lat.device
        gy, lat = _dense32(gy, lat)
        b, l, k = lat.shape
        n = ws[0].shape[0]
        slot_t = _ptr_table_ints(slot, dev)
        glat = _grouped_dgrad(gy, ws, slot, wscale, l) if need[0] else None
        grads_w, grads_b = [None] * g, [None] * g
        if any(need[4:]):
            # every layer's results go straight to the parameter's slice of the flat gradient store where that is free
            # (_grad_dest), into a stacked buffer otherwise: one launch either way, and no per-layer accumulation copy
            gw = gb = None
            dw, db, direct = [], [], 0
            for j in range(g):
                tw = _grad_dest(ws[j]) if need[4 + j] else None
                tb = _grad_dest(bs[j]) if need[4 + g + j] else None
                direct += (tw is not None) + (tb is not None)
                if tw is None:
                    if gw is None:
                        gw = torch.empty((g, n, k), dtype=torch.float32, device=dev)
                    tw = gw[j]
                if tb is None:
                    if gb is None:
                        gb = torch.empty((g, n), dtype=torch.float32, device=dev)
                    tb = gb[j]
                dw.append(tw)
                db.append(tb)
            if direct == 0:
                # no destination in the flat store at all (no armed reducer): the stacked form, no pointer table to ship
                with _lib.on_device(dev):
                    code = _lib.lib().msg_linear_grouped_wgrad(gy.data_ptr(), lat.data_ptr(), slot_t.data_ptr(), gw.data_ptr(),
                                                               gb.data_ptr(), g, b, n, k, l, wscale, bias_scale,
                                                               _lib.stream_of(dev))
                _lib.check(code, "msg_linear_grouped_wgrad")
            else:
                if gw is None and gb is None:
                    tbl = _ptr_table(dw + db, dev).view(2, g)     # (flat-store addresses: stable, the table is cached)
                else:
                    tbl = torch.tensor([[t.data_ptr() for t in dw], [t.data_ptr() for t in db]], dtype=torch.int64).to(dev)
                with _lib.on_device(dev):
                    code = _lib.lib().msg_linear_grouped_wgrad_ptrs(gy.data_ptr(), lat.data_ptr(), slot_t.data_ptr(),
                                                                    tbl[0].data_ptr(), tbl[1].data_ptr(), g, b, n, k, l,
                                                                    wscale, bias_scale, _lib.stream_of(dev))
                _lib.check(code, "msg_linear_grouped_wgrad_ptrs")
            grads_w = [dw[j] if need[4 + j] else None for j in range(g)]
            grads_b = [db[j] if need[4 + g + j] else None for j in range(g)]
        return (glat, None, None, None, *grads_w, *grads_b)


def _grouped_dgrad(gy, ws, slot, wscale, n_slots):
    """glat [B, L, K] = for every latent slot the ordered sum over the layers that read it of wscale * gy[g] @ W_g."""
    dev = gy.device
    g = len(slot)
    (gy,) = _dense32(gy)
    b, n, k = gy.shape[1], ws[0].shape[0], ws[0].shape[1]
    gx = torch.empty((g + 1, b, k), dtype=torch.float32, device=dev)
    gx[g].zero_()                                   # (the row the padding entries of the gather table point at)
    with _lib.on_device(dev):
        code = _lib.lib().msg_linear_grouped_dgrad(gy.data_ptr(), _ptr_table(ws, dev).data_ptr(), gx.data_ptr(),
                                                   g, b, n, k, wscale, _lib.stream_of(dev))
    _lib.check(code, "msg_linear_grouped_dgrad")
    # layers that read the same latent slot: gather + ordered sum (index_add_ adds with float atomics, i.e. in an
    # order that changes from run to run)
    return gx[_slot_gather_table(slot, n_slots, dev)].sum(dim=1).transpose(0, 1)


class _GroupedLinD(Function):
    """The latent's gradient of _GroupedLinear as an op of its own, differentiable once more: for a cotangent v [B, L, K] of
    glat,  d/d gy[g] = wscale * v[:, slot g] @ W_g^T  (the grouped forward without bias)  and  d/d W_g = wscale * gy[g]^T @
    v[:, slot g]  (the grouped weight gradient with v in the latent's place)."""

    @staticmethod
    def forward(ctx, gy, slot, wscale, n_slots, *ws):
        _lib.require_gpu(gy, *ws)
        ctx.save_for_backward(gy, *ws)
        ctx.cfg = (tuple(slot), float(wscale), int(n_slots))
        return _grouped_dgrad(gy, ws, slot, wscale, n_slots)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        gy, *ws = ctx.saved_tensors
        slot, wscale, l = ctx.cfg
        g, dev = len(slot), gy.device
        gy, v = _dense32(gy, v)
        b, n, k = gy.shape[1], ws[0].shape[0], ws[0].shape[1]
        slot_t = _ptr_table_ints(slot, dev)
        d_gy, d_ws = None, [None] * g
        with _lib.on_device(dev):
            if ctx.needs_input_grad[0]:
                d_gy = torch.empty((g, b, n), dtype=torch.float32, device=dev)
                code = _lib.lib().msg_linear_grouped_fprop(v.data_ptr(), slot_t.data_ptr(), _ptr_table(ws, dev).data_ptr(), None,
                                                           d_gy.data_ptr(), g, b, n, k, l, wscale, 1.0, _lib.stream_of(dev))
                _lib.check(code, "msg_linear_grouped_fprop")
            if any(ctx.needs_input_grad[4:]):
                gw = torch.empty((g, n, k), dtype=torch.float32, device=dev)
                code = _lib.lib().msg_linear_grouped_wgrad(gy.data_ptr(), v.data_ptr(), slot_t.data_ptr(), gw.data_ptr(), None, g, b,
                                                           n, k, l, wscale, 1.0, _lib.stream_of(dev))
                _lib.check(code, "msg_linear_grouped_wgrad")
                d_ws = [gw[j] if ctx.needs_input_grad[4 + j] else None for j in range(g)]
        return (d_gy, None, None, None, *d_ws)


_INT_TABLES: dict = {}
_GATHER_TABLES: dict = {}


def _slot_gather_table(slot, n_slots, dev):
    """[n_slots, max layers per slot] int64: row l lists the layers whose latent slot is l, padded with len(slot) (the index
    of a zero row appended to what is gathered)."""
    key = (dev.index, tuple(int(v) for v in slot), n_slots)
    hit = _GATHER_TABLES.get(key)
    if hit is None:
        if len(_GATHER_TABLES) > 64:
            _GATHER_TABLES.clear()
        rows = [[j for j, v in enumerate(key[1]) if v == l] for l in range(n_slots)]
        width = max(1, max(len(r) for r in rows))
        hit = _GATHER_TABLES[key] = torch.tensor([r + [len(slot)] * (width - len(r)) for r in rows], dtype=torch.int64,
                                                 device=dev)
    return hit


def _ptr_table_ints(values, dev):
    key = (dev.index, tuple(int(v) for v in values))
    hit = _INT_TABLES.get(key)
    if hit is None:
        if len(_INT_TABLES) > 64:
            _INT_TABLES.clear()
        hit = _INT_TABLES[key] = torch.tensor(key[1], dtype=torch.int32, device=dev)
    return hit


def grouped_linear(latent, slots, weights, biases, wscale, bias_scale):
    """[G, B, N]: layer j applied to latent[:, slots[j]] -- all G style affines of the generator in one launch."""
    return _GroupedLinear.apply(latent, tuple(int(v) for v in slots), float(wscale), float(bias_scale), *weights, *biases)


def demod_coefficients(weight, style, scale):
    """d[b,o] = rsqrt(scale^2 * sum_i s[b,i]^2 * sum_k W[o,i,k]^2 + 1e-8)   (generator.py:384-388, refactored)."""
    wsq = weight[0].square().sum(dim=(2, 3))
    return torch.rsqrt((scale * scale) * (style.square() @ wsq.t()) + 1e-8)


def _modulated_composite(x, weight, style, demodulate, upsample):
    """Differentiable-to-any-order formulation: torch ops build w_b = d * scale * W * s, then one batched contraction."""
    _, out_c, in_c, kh, kw = weight.shape
    scale = math.sqrt(2.0) / math.sqrt(in_c * kh * kw)
    # ONE pass over the [B, O, I, kh, kw] per-sample weights (75 MB a layer at 512 channels): the per-(sample, o, i)
    # coefficient is formed first, on B*O*I elements -- (weight * scale) * style * d took two to three such passes here and
    # twice that in every differentiation of the graph
    coef = style[:, None, :] * scale
    if demodulate:
        coef = coef * demod_coefficients(weight, style, scale)[:, :, None]
    wmod = weight * coef[:, :, :, None, None]
    g = Geometry("up2" if upsample else "conv", kh, kw, 1, kh // 2, x.shape[2:], True)
    return _ConvF.apply(x, wmod, None, g)


def _scale_rows_cols(base, rowscale, colscale, out, gain):
    b, r, t, ck = out.shape
    c = base.shape[-1]
    dev = base.device
    with _lib.on_device(dev):
        code = _lib.lib().msg_scale_rows_cols(base.data_ptr(), _lib.ptr(rowscale), _lib.ptr(colscale), out.data_ptr(),
                                              _lib.dtype_code(out), b, r, t, c, ck, float(gain), _lib.stream_of(dev))
    _lib.check(code, "msg_scale_rows_cols")
    return out


_MODCONV_BWD_BATCH = 16


def _dgrad_image_base(weight, w3, upsample, kind):
    """fp32 base of the data-gradient weight image, [I][taps (flipped for convs)][O]."""
    img = _param_images(weight, torch.float32, 1.0, kind, modulation=True)
    return img["d"][0] if img is not None else _cached(weight, "md" + kind, torch.float32, 1.0, lambda: (
        (w3 if upsample else w3.flip(-1)).permute(1, 2, 0).contiguous(), 0))[0]


def _fwd_image_base(weight, w3, upsample, kind):
    """fp32 base of the forward weight image: conv -> [O][taps][I]; up2 -> rows n = q*O + o, one tap."""
    o, i, t = w3.shape
    img = _param_images(weight, torch.float32, 1.0, kind, modulation=True)
    return img["f"][0] if img is not None else _cached(weight, "mb" + kind, torch.float32, 1.0, lambda: (
        (w3.permute(2, 0, 1).reshape(t * o, 1, i) if upsample else w3.transpose(1, 2)).contiguous(), 0))[0]


def _tap_square_sums(weight, w3, kind):
    """wsq[o][i] = sum over taps of W^2 (cached with the weight)."""
    img = _param_images(weight, torch.float32, 1.0, kind, modulation=True)
    return img["wsq"] if img is not None else \
        _cached(weight, "wsq", torch.float32, 1.0, lambda: (w3.square().sum(dim=2).contiguous(), 0))[0]


def _dgrad_modconv(gy, wd, okp, o, i, kh, kw, upsample, g, act_bwd=None, sign_map=None):
    """Data-gradient contraction of the modulated conv with a per-sample data-gradient weight image; act_bwd: the ActHandle
    of the layer that produced the conv's input (its activation backward then runs in this launch's epilogue)."""
    if act_bwd is not None and act_bwd.armed and not upsample and kh == 3 and kw == 3:
        out = _launch_dgrad_act_backward(gy, wd, okp, i, kh, kw, True, o, act_bwd, mode=g.mode, sign_map=sign_map)
        if out is not None:
            return out
    if upsample:
        return _launch_fprop(gy, wd, okp, None, i, g.x_hw, 2, 2, 2, 0, 1, False, True, o, mode=g.mode)
    return _launch_fprop(gy, wd, okp, None, i, g.x_hw, kh, kw, 1, kh - 1 - kh // 2, 1, False, True, o, mode=g.mode)


def _fprop_modconv(x, wk, ck, o, i, kh, kw, upsample, g, add_to=None):
    """Forward contraction of the modulated conv with a per-sample forward weight image; `add_to`: a map shaped like the
    result that is added to it (in the contraction's epilogue where the kernel has one: the plain stride-1 convs)."""
    if upsample:
        y = _launch_fprop(x, wk, ck, None, 4 * o, g.x_hw, 1, 1, 1, 0, 1, True, True, i, mode=g.mode)
        return y if add_to is None else y + add_to
    return _launch_fprop(x, wk, ck, None, o, g.y_hw, kh, kw, 1, kh // 2, 1, False, True, i,
                         residual=None if add_to is None else (add_to, 1.0), mode=g.mode)


def _wgrad_modconv(gy, x, o, i, kh, kw, upsample, g):
    """Per-sample weight gradient in the kernel layout [B][O][taps][ldg] -> (gwk, ldg)."""
    if upsample:
        return _launch_wgrad(gy, x, o, i, 2, 2, 1, 0, True, True, g.x_hw, raw=True, mode=g.mode)
    return _launch_wgrad(gy, x, o, i, kh, kw, 1, kh // 2, False, True, None, raw=True, mode=g.mode)


def _fold_weight_gradient(gwk, ldg, w3, s, dd, scale, style_dtype, cotangent=None, dest=None):
    """Per-sample weight gradients -> (dL/dW [1,O,I,kh*kw flat], dL/ds) through msg_modulate_backward; with `cotangent` (v,
    the cotangent of a FIRST backward's style gradient) the second-order terms of msg_modulate_backward2 instead."""
    b = gwk.shape[0]
    o, i, t = w3.shape
    dev = gwk.device
    og = 2 if o >= 512 or o < 256 else 1           # >= 256 workgroups; 512 outputs: 256 partial rows of the style gradient, not 512
    groups = (o + og - 1) // og
    out = _grad_dest(dest) if cotangent is None else None     # (the parameter's slice of the flat gradient store)
    gw3 = out.view(o, i, t) if out is not None else torch.empty((o, i, t), dtype=torch.float32, device=dev)
    gs_part = _lib.scratch_ptr(groups * b * i, dev)           # [groups][B][I] partials: live until the row sum two lines below
    gs = torch.empty((b, i), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        st = _lib.stream_of(dev)
        if cotangent is None:
            code = _lib.lib().msg_modulate_backward(gwk.data_ptr(), w3.data_ptr(), s.data_ptr(), _lib.ptr(dd),
                                                    gw3.data_ptr(), gs_part, b, o, i, t, ldg, og, scale, st)
        else:
            code = _lib.lib().msg_modulate_backward2(gwk.data_ptr(), w3.data_ptr(), s.data_ptr(), _lib.ptr(dd),
                                                     cotangent.data_ptr(), gw3.data_ptr(), gs_part, b, o, i, t,
                                                     ldg, og, scale, st)
        _lib.check(code, "msg_modulate_backward" + ("2" if cotangent is not None else ""))
        code = _lib.lib().msg_sum_rows(gs_part, gs.data_ptr(), groups, b * i, st)
    _lib.check(code, "msg_sum_rows")
    return (out if out is not None else gw3), (gs if style_dtype == torch.float32 else gs.to(style_dtype))


def _modconv_backward(x, weight, style, d, gy, demodulate, upsample, g, scale, need, keep_gwk=False, direct=False,
                      act_bwd=None):
    """First-order backward of the modulated convolution for at most 16 samples: data gradient with re-laid per-sample
    weights, per-sample weight gradient, and the kernel that folds it into dL/dW and dL/ds (see _ModulatedConv)."""
    _, o, i, kh, kw = weight.shape
    b, t = x.shape[0], kh * kw
    dev = x.device
    w3 = weight.detach().reshape(o, i, t)
    s = style.detach().float().contiguous()
    dd = d if demodulate else None
    esz = 2 if gy.dtype == torch.bfloat16 else 4
    gx = None
    if need[0]:
        okp = _round_up(o, 128 // esz)
        wd = torch.empty((b, i, t, okp), dtype=gy.dtype, device=dev)
        _scale_rows_cols(_dgrad_image_base(weight, w3, upsample, g.kind), s, dd, wd, scale)
        gx = _dgrad_modconv(gy, wd, okp, o, i, kh, kw, upsample, g, act_bwd=act_bwd, sign_map=x)
    gw = gs = gwk = None
    if need[1] or need[2]:
        gwk, ldg = _wgrad_modconv(gy, x, o, i, kh, kw, upsample, g)
        gw3, gs = _fold_weight_gradient(gwk, ldg, w3, s, dd, scale, style.dtype, dest=weight if direct else None)
        gw = gw3.reshape(1, o, i, kh, kw)
    return (gx, gw, gs, gwk) if keep_gwk else (gx, gw, gs)


_NATIVE_SECOND_ORDER = True       # tests flip it to check the native second-order node against the composite torch-op graph


def _modconv_second_order_composite(gy, x, weight, style, demodulate, upsample, a, cot_w, cot_s, want):
    """Gradients of L2 = <a, gx> + <cot_w, gW> + <cot_s, gs> with respect to (gy, x, weight, style), where (gx, gW, gs) is the
    first backward of the modulated conv: the general fallback, by differentiating the composite torch-op formulation twice."""
    with torch.enable_grad():
        leaves = [t.detach().requires_grad_(True) for t in (gy, x, weight, style)]
        y2 = _modulated_composite(leaves[1], leaves[2], leaves[3], demodulate, upsample)
        first = torch.autograd.grad(y2, leaves[1:], leaves[0], create_graph=True, allow_unused=True)
        terms = [(c.to(f.dtype) * f).sum() for c, f in zip((a, cot_w, cot_s), first) if c is not None and f is not None]
        if not terms:
            return (None, None, None, None)
        grads = torch.autograd.grad(sum(terms), leaves, allow_unused=True)
    return tuple(gr if w else None for gr, w in zip(grads, want))


class _ModConvGrad(Function):
    """The FIRST backward of the modulated convolution as a differentiable op, (gy, x, W, s) -> (gx, gW, gs), so that the
    second-order pass of the path-length regulariser (multi_stylegan_generator.py:193-200 through :384-411) runs on the
    same native pieces as the first-order step instead of a composite graph of torch elementwise ops over [B,O,I,kh,kw]
    tensors (16 GB of elementwise traffic and one recomputed forward contraction per regularised iteration):

        d<a,gx>/d(gy, W, s):  F(a, w),  MB(G(gy, a))                     w = per-sample weights, MB = msg_modulate_backward
        d<v,gs>/d(gy, x):     F(x, dw), D(gy, dw)                        dw = derivative of w along v (msg_scale_rows_cols2)
        d<v,gs>/d(W, s):      msg_modulate_backward2 on the saved per-sample weight gradient

    A cotangent for gW (nothing in the training step produces one), more than 16 samples or shapes outside the fold kernels'
    limits take the composite fallback."""

    @staticmethod
    def forward(ctx, gy, x, weight, style, d, demodulate, upsample, g, scale, need):
        gx, gw, gs, gwk = _modconv_backward(x, weight, style, d, gy, demodulate, upsample, g, scale, need, keep_gwk=True)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(gy, x, weight, style, d, gwk)
        ctx.cfg = (demodulate, upsample, g, scale)
        return gx, gw, gs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, a, cot_w, cot_s):
        gy, x, weight, style, d, gwk = ctx.saved_tensors
        demodulate, upsample, g, scale = ctx.cfg
        want = ctx.needs_input_grad[:4]
        _, o, i, kh, kw = weight.shape
        b, t = x.shape[0], kh * kw
        tail = (None,) * 6
        if a is None and cot_w is None and cot_s is None:
            return (None, None, None, None) + tail
        if cot_w is not None or i % 4 or i > 512 or t not in (1, 4, 9) or b > _MODCONV_BWD_BATCH or \
                (cot_s is not None and gwk is None):
            return _modconv_second_order_composite(gy, x, weight, style, demodulate, upsample, a, cot_w, cot_s, want) + tail
        dev = x.device
        w3 = weight.detach().reshape(o, i, t)
        s = style.detach().float().contiguous()
        dd = d if demodulate else None
        esz = 2 if gy.dtype == torch.bfloat16 else 4
        ck, okp = _round_up(i, 128 // esz), _round_up(o, 128 // esz)
        rows = t * o if upsample else o
        ggy = gx2 = gw3 = gs2 = None
        if a is not None:
            a = a.to(x.dtype)
            if want[0]:                                   # <a, D(gy, w)> is F(a, w) paired with gy
                wk = torch.empty((b, rows, 1 if upsample else t, ck), dtype=x.dtype, device=dev)
                base = _fwd_image_base(weight, w3, upsample, g.kind)
                if demodulate:
                    _scale_rows_cols(base, d.repeat(1, t) if upsample else d, s, wk, scale)
                else:
                    _scale_rows_cols(base, None, s, wk, scale)
                ggy = _fprop_modconv(a, wk, ck, o, i, kh, kw, upsample, g)
            if want[2] or want[3]:                        # ... and G(gy, a) paired with w -> (W, s) through the fold
                gwk_a, ldg = _wgrad_modconv(gy, a, o, i, kh, kw, upsample, g)
                gw3, gs2 = _fold_weight_gradient(gwk_a, ldg, w3, s, dd, scale, style.dtype)
        if cot_s is not None:
            v = cot_s.detach().float().contiguous()
            if want[0] or want[1]:
                # dw = derivative of the weight set along v, in both kernel layouts
                hf = torch.empty((b, rows, 1 if upsample else t, ck), dtype=x.dtype, device=dev)
                hd = torch.empty((b, i, t, okp), dtype=gy.dtype, device=dev)
                base_f = _fwd_image_base(weight, w3, upsample, g.kind)
                base_d = _dgrad_image_base(weight, w3, upsample, g.kind)
                if demodulate:
                    wsq = _tap_square_sums(weight, w3, g.kind)
                    ddir = -(scale * scale) * d.pow(3) * linear(s * v, wsq)          # [B, O]
                    rows_d, rows_dd = (d.repeat(1, t), ddir.repeat(1, t)) if upsample else (d, ddir)
                    with _lib.on_device(dev):
                        lib, st = _lib.lib(), _lib.stream_of(dev)
                        code = lib.msg_scale_rows_cols2(base_f.data_ptr(), rows_d.contiguous().data_ptr(), v.data_ptr(),
                                                        rows_dd.contiguous().data_ptr(), s.data_ptr(), hf.data_ptr(),
                                                        _lib.dtype_code(hf), b, rows, 1 if upsample else t, i, ck, scale, st)
                        _lib.check(code, "msg_scale_rows_cols2")
                        code = lib.msg_scale_rows_cols2(base_d.data_ptr(), v.data_ptr(), d.data_ptr(), s.data_ptr(),
                                                        ddir.contiguous().data_ptr(), hd.data_ptr(), _lib.dtype_code(hd), b, i, t,
                                                        o, okp, scale, st)
                        _lib.check(code, "msg_scale_rows_cols2")
                else:
                    _scale_rows_cols(base_f, None, v, hf, scale)
                    _scale_rows_cols(base_d, v, None, hd, scale)
                if want[0]:                               # <dw, G(gy, x)> is F(x, dw) paired with gy ...
                    ggy = _fprop_modconv(x, hf, ck, o, i, kh, kw, upsample, g, add_to=ggy)
                if want[1]:                               # ... and D(gy, dw) paired with x
                    gx2 = _dgrad_modconv(gy, hd, okp, o, i, kh, kw, upsample, g)
            if want[2] or want[3]:
                gw3b, gs2b = _fold_weight_gradient(gwk, gwk.shape[-1], w3, s, dd, scale, style.dtype, cotangent=v)
                gw3 = gw3b if gw3 is None else gw3 + gw3b
                gs2 = gs2b if gs2 is None else gs2 + gs2b
        gw = gw3.reshape(1, o, i, kh, kw) if (gw3 is not None and want[2]) else None
        return (ggy, gx2, gw, gs2 if want[3] else None) + tail


class _ModulatedConv(Function):
    """The dual-styled modulated / demodulated convolution with fused weight handling (csrc/modulate.hip):
    forward  = demod coefficients (wave-shuffle reduction) -> per-sample weights written straight in the kernel
               layout -> batched MFMA contraction;
    backward = data gradient with re-laid per-sample weights, per-sample weight gradient (TN kernel), then ONE kernel
               that folds it into dL/dW and dL/ds including the derivative of the demodulation norm.
    When a higher-order graph is requested (path-length regularisation) the backward re-derives itself from the
    composite formulation, which is differentiable to any order."""

    @staticmethod
    def forward(ctx, x, weight, style, demodulate, upsample, act_bias=None, noise=None, noise_w=None, alpha=0.2,
                act_scale=1.0, fuse_act=False, head_slot=None, input_act=None):
        dev = _lib.require_gpu(x, weight, style)
        ctx.input_act = input_act if input_act is not None and input_act.armed else None     # (ActHandle of x's producer)
        # head_slot (HeadGradSlot): on a styled layer WITH its activation, the slot its level's image head fills in backward;
        # on the head itself (no activation), the slot to fill.  The producer may then see no incoming gradient at all.
        ctx.head_slot = head_slot
        if head_slot is not None and fuse_act:
            ctx.set_materialize_grads(False)
            head_slot.armed = True
        _, o, i, kh, kw = weight.shape
        b, t = x.shape[0], kh * kw
        scale = math.sqrt(2.0) / math.sqrt(i * t)
        w3 = weight.detach().reshape(o, i, t)
        s = style.detach().float().contiguous()
        d = None
        esz = 2 if x.dtype == torch.bfloat16 else 4
        ck = _round_up(i, 128 // esz)
        kind = "up2" if upsample else "conv"
        # base [R][T][C]: conv -> [O][taps][I];  up2 -> rows n = q*O + o, one tap
        img = _param_images(weight, torch.float32, 1.0, kind, modulation=True)
        base = img["f"][0] if img is not None else _cached(weight, "mb" + kind, torch.float32, 1.0, lambda: (
            (w3.permute(2, 0, 1).reshape(t * o, 1, i) if upsample else w3.transpose(1, 2)).contiguous(), 0))[0]
        rows = t * o if upsample else o
        wk = torch.empty((b, rows, 1 if upsample else t, ck), dtype=x.dtype, device=dev)
        if demodulate:
            # demodulation coefficients + weight set in one launch; sum_t W^2 is cached with the weight
            wsq = img["wsq"] if img is not None else \
                _cached(weight, "wsq", torch.float32, 1.0, lambda: (w3.square().sum(dim=2).contiguous(), 0))[0]
            d = torch.empty((b, o), dtype=torch.float32, device=dev)
            with _lib.on_device(dev):
                code = _lib.lib().msg_modulate_weights(base.data_ptr(), wsq.data_ptr(), s.data_ptr(), wk.data_ptr(),
                                                       d.data_ptr(), _lib.dtype_code(wk), b, rows, o,
                                                       1 if upsample else t, i, ck, scale, 1e-8, _lib.stream_of(dev))
            _lib.check(code, "msg_modulate_weights")
        else:
            _scale_rows_cols(base, None, s, wk, scale)
        g = Geometry(kind, kh, kw, 1, kh // 2, x.shape[2:], True)
        act = None
        holder = [] if any(ctx.needs_input_grad) else None   # (a forward that no backward follows writes no sign bytes)
        if fuse_act:
            assert not upsample, "the upsampling layers blur before their activation"
            act = (*_act_operands(act_bias, noise, noise_w, (b, o, *g.y_hw)), alpha, act_scale, holder)
        if upsample:
            y = _launch_fprop(x, wk, ck, None, 4 * o, g.x_hw, 1, 1, 1, 0, 1, True, True, i, mode=g.mode)
        else:
            y = _launch_fprop(x, wk, ck, None, o, g.y_hw, kh, kw, 1, kh // 2, 1, False, True, i, act=act, mode=g.mode)
        ctx.save_for_backward(x, weight, style, d if d is not None else torch.empty(0, device=dev),
                              y if fuse_act else None, noise if fuse_act else None)
        ctx.cfg = (demodulate, upsample, g, scale)
        ctx.act = (alpha, act_scale, act_bias is not None, noise is not None,
                   None if noise_w is None else noise_w.shape) if fuse_act else None
        ctx.bias_param = act_bias
        ctx.mask = holder[0] if holder else None            # sign bytes of y, when the forward kernel wrote them
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, style, d, y_act, noise = ctx.saved_tensors
        demodulate, upsample, g, scale = ctx.cfg
        _, o, i, kh, kw = weight.shape
        b, t = x.shape[0], kh * kw
        need = list(ctx.needs_input_grad)
        need[1], need[2] = _consumed(ctx, 1, 1), _consumed(ctx, 2, 2)       # (weight, style: see _consumed)
        gb = gnw = None
        slot = ctx.head_slot
        if ctx.act is not None:
            # activation stage first (slope from the sign of the saved OUTPUT); a differentiable Function, so the
            # second-order graph of path-length regularisation runs through it exactly as in the two-pass form
            from .op_static.fused_act import FusedLeakyReLUFunctionBackward, act_backward_with_head
            alpha, act_scale, has_bias, has_noise, nw_shape = ctx.act
            head, head_dgrad = slot.take() if slot is not None else (None, None)
            fused = None
            if head is not None:
                # the level's image head left its gradient here instead of writing it as a map: formed inside the pass
                fused = act_backward_with_head(gy, head, y_act.shape, noise if has_noise else None,
                                               ctx.bias_param if has_bias else None, has_bias, alpha, act_scale, ctx.mask)
                if fused is None:
                    gh = head_dgrad()
                    gy = gh if gy is None else gy + gh
            if fused is not None:
                gy, gb, gnw = fused
            elif gy is None:
                return (None,) * 13
            else:
                gy, gb, gnw = _derive(FusedLeakyReLUFunctionBackward, gy, y_act, noise if has_noise else None,
                                      ctx.bias_param if has_bias else False, alpha, act_scale, ctx.mask)
            gb = gb if has_bias and need[5] else None
            gnw = gnw.reshape(nw_shape) if has_noise and need[7] else None
        tail = (None, None, gb, None, gnw, None, None, None, None, None)
        if ctx.act is None and slot is not None and slot.armed and need[0] and not torch.is_grad_enabled() and \
                fused_head_ok(gy, weight, b, demodulate, upsample):
            # this IS the head: weight / style gradients as always, the data gradient handed to the producer's activation backward
            gyc, wt, st = gy, weight, style
            slot.head = (gyc, weight.detach().reshape(o, i).float().contiguous(), style.detach().float().contiguous(), scale)
            slot.dgrad = lambda: _modconv_backward(x, wt, st, d, gyc, demodulate, upsample, g, scale,
                                                   [True, False, False])[0]
            need[0] = False
        fused_ok = i <= 512 and t <= 9
        if torch.is_grad_enabled() and fused_ok and _NATIVE_SECOND_ORDER and b <= _MODCONV_BWD_BATCH and x.is_cuda:
            # create_graph=True (the path-length pass): the first backward as ONE differentiable node on the native kernels
            gx, gw, gs = _derive(_ModConvGrad, gy, x, weight, style, d if demodulate else None, demodulate, upsample, g, scale,
                                            tuple(need[:3]))
            return (gx, gw, gs) + tail
        if torch.is_grad_enabled() or not fused_ok:
            # higher-order request (create_graph=True): differentiate the composite formulation instead
            with torch.enable_grad():
                ins = [v for v, n in zip((x, weight, style), need[:3]) if n]
                y2 = _modulated_composite(x, weight, style, demodulate, upsample)
                grads = list(torch.autograd.grad(y2, ins, gy, create_graph=torch.is_grad_enabled(), allow_unused=True))
            out = [grads.pop(0) if n else None for n in need[:3]]
            return (out[0], out[1], out[2]) + tail
        # msg_modulate_backward keeps one sample's partial sums per unrolled register slot: 16 samples per launch.  Larger
        # batches go through in chunks of 16 (per-sample results concatenated, the weight gradient summed).
        if b > _MODCONV_BWD_BATCH:
            gxs, gws, gss = [], [], []
            for lo in range(0, b, _MODCONV_BWD_BATCH):
                hi = min(b, lo + _MODCONV_BWD_BATCH)
                cgx, cgw, cgs = _modconv_backward(x[lo:hi], weight, style[lo:hi], d[lo:hi] if demodulate else d,
                                                  gy[lo:hi], demodulate, upsample, g, scale, need)
                gxs.append(cgx); gws.append(cgw); gss.append(cgs)
            gx = torch.cat(gxs) if need[0] else None
            gw = torch.stack(gws).sum(dim=0) if gws[0] is not None else None
            gs = torch.cat(gss) if gss[0] is not None else None
            return (gx, gw, gs) + tail
        gx, gw, gs = _modconv_backward(x, weight, style, d, gy, demodulate, upsample, g, scale, need, direct=True,
                                       act_bwd=ctx.input_act)
        return (gx, gw, gs) + tail


def fused_head_ok(gy, weight, b, demodulate, upsample) -> bool:
    """Whether a modulated conv is an image head whose data gradient the producer's activation backward can form itself
    (HeadGradSlot): 1x1, no demodulation, at most 8 planes, bf16 gradient, one modconv-backward batch."""
    _, o, _, kh, kw = weight.shape
    return gy is not None and gy.is_cuda and gy.dtype == torch.bfloat16 and kh == 1 and kw == 1 and o <= 8 and \
        not demodulate and not upsample and b <= _MODCONV_BWD_BATCH and HEAD_GRAD_FUSION


HEAD_GRAD_FUSION = bool(int(os.environ.get("MSG_HEAD_GRAD_FUSION", "1")))   # 0 / False: the head writes its data gradient as a map (A/B; tests compare the two forms)


def modulated_conv2d_bias_act(x, weight, style, demodulate, act_bias, noise, noise_weight, negative_slope=0.2,
                              scale=1.0, head_slot=None, input_act=None):
    """modulated_conv2d (no upsampling) -> noise injection -> bias -> leaky ReLU, the activation stage fused into the
    contraction's epilogue (multi_stylegan_generator.py:267-292 + 384-411 in one pass over the output map).
    head_slot: the HeadGradSlot the level's image head fills in backward (see there); input_act: the ActHandle of the layer
    that produced x, when this conv is x's only consumer."""
    return _ModulatedConv.apply(x, weight, style, bool(demodulate), False, act_bias, noise, noise_weight,
                                float(negative_slope), float(scale), True, head_slot, input_act)


def modulated_conv2d(x, weight, style, demodulate, upsample, head_slot=None):
    """x [B,I,H,W]; weight [1,O,I,kh,kw] fp32; style [B,I] fp32 -> conv result (before any blur).

    One weight set per sample, w_b = d[b,o] * scale * W[o,i,k] * s[b,i] (multi_stylegan_generator.py:384-388), and
    one batched contraction (grid.z = sample) instead of the reference's groups=batch library conv.
    head_slot: this conv is an image head and `x` the output of the styled layer that holds the same HeadGradSlot."""
    if head_slot is None:
        return _ModulatedConv.apply(x, weight, style, bool(demodulate), bool(upsample))
    return _ModulatedConv.apply(x, weight, style, bool(demodulate), bool(upsample), None, None, None, 0.2, 1.0, False,
                                head_slot)
