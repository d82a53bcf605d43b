"""ctypes binding of libmsg_hip.so (C ABI: include/msg_hip.h).

There is no CPU fallback: if the shared library is missing or a call reports
an error this raises.  Tensors are passed as raw device pointers together with
the HIP stream torch is currently recording on, so launches interleave
correctly with torch's own kernels.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSG_LIB_VARIANT=<tag> loads libmsg_hip_<tag>.so instead: an experimental or diagnostic build kept beside the library for
# same-box A/B timing (tools/microbench.py, tools/row3_stamps.py); never set in tests or in the benchmark
LIB_PATH = os.path.join(_HERE, "libmsg_hip" + ("_" + os.environ["MSG_LIB_VARIANT"] if os.environ.get("MSG_LIB_VARIANT") else "")
                        + ".so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "msg_hip.h")

MSG_F32, MSG_BF16, MSG_F16, MSG_F64 = 0, 1, 2, 3
ABI_VERSION = 5          # MSG_ABI_VERSION of include/msg_hip.h this binding was written against (checked at load time)
_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_longlong, _c.c_float

_SIGNATURES = {
    "msg_abi_version": (_I, []),
    "msg_build_arch": (_c.c_char_p, []),
    "msg_strerror": (_c.c_char_p, [_I]),
    "msg_upfirdn2d": (_I, [_P, _P, _P, _I] + [_I] * 14 + [_P]),
    "msg_upfirdn2d_pitched": (_I, [_P, _P, _P, _I] + [_I] * 15 + [_P]),
    "msg_upfirdn2d_pitched2": (_I, [_P, _P, _P, _I] + [_I] * 16 + [_P]),
    "msg_upfirdn2d_separable": (_I, [_P, _P, _P, _P, _I] + [_I] * 10 + [_P]),
    "msg_upfirdn2d_separable_act": (_I, [_P, _P, _P, _P, _I] + [_I] * 10 + [_P, _P, _P, _I, _F, _F, _P]),
    "msg_upfirdn2d_separable_act_mask": (_I, [_P, _P, _P, _P, _I] + [_I] * 10 + [_P, _P, _P, _I, _F, _F, _P, _P]),
    "msg_fused_bias_act": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "msg_bias_act_backward": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _P, _P, _I, _I, _F, _F, _P, _L, _P]),
    "msg_channel_sums": (_I, [_P, _P, _I, _L, _I, _P, _L, _P]),
    "msg_bias_act_backward_mask": (_I, [_P, _P, _I, _I, _P, _I, _L, _I, _P, _P, _P, _I, _I, _F, _F, _P, _L, _P]),
    "msg_bias_act_backward_mask_head": (_I, [_P, _P, _P, _P, _F, _I, _P, _I, _I, _P, _I, _L, _I, _P, _P, _P, _I, _I, _F, _F,
                                             _P, _L, _P]),
    "msg_bias_act_backward_workspace": (_L, [_L, _I, _I, _I]),
    "msg_conv2d_fprop": (_I, [_P, _P, _P, _P, _I] + [_I] * 15 + [_L, _P]),
    "msg_conv2d_fprop_act": (_I, [_P, _P, _P, _I] + [_I] * 13 + [_L, _P, _P, _P, _I, _F, _F, _P]),
    "msg_conv2d_fprop_act_mask": (_I, [_P, _P, _P, _I] + [_I] * 13 + [_L, _P, _P, _P, _I, _F, _F, _P, _P]),
    "msg_conv2d_fprop_residual": (_I, [_P, _P, _P, _I] + [_I] * 13 + [_L, _P, _I, _F, _P]),
    "msg_conv2d_fprop_act_backward_workspace": (_L, [_I] * 11 + [_L, _I]),
    "msg_sum_rows": (_I, [_P, _P, _L, _I, _P]),
    "msg_act_pointwise_head": (_I, [_P, _P, _P, _P, _I, _L, _I, _F, _F, _F, _P]),
    "msg_act_pointwise_head_backward_workspace": (_L, [_L, _I]),
    "msg_act_pointwise_head_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _F, _F, _F, _P, _L, _P]),
    "msg_conv2d_fprop_act_backward": (_I, [_P, _P, _P, _I] + [_I] * 13 + [_L, _P, _I, _P, _I, _I, _P, _I, _F, _F, _P, _P, _I, _P,
                                           _P, _L, _P]),
    "msg_conv2d_wgrad": (_I, [_P, _P, _P, _I] + [_I] * 18 + [_F, _P, _L, _P]),
    "msg_conv2d_wgrad_workspace": (_L, [_I] * 18),
    "msg_demod_coeff": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "msg_scale_rows_cols": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "msg_modulate_weights": (_I, [_P] * 5 + [_I] * 7 + [_F, _F, _P]),
    "msg_modulate_backward": (_I, [_P] * 6 + [_I] * 6 + [_F, _P]),
    "msg_modulate_backward2": (_I, [_P] * 7 + [_I] * 6 + [_F, _P]),
    "msg_scale_rows_cols2": (_I, [_P] * 6 + [_I] * 6 + [_F, _P]),
    "msg_relayout_weight": (_I, [_P, _P, _P, _P, _I] + [_I] * 7 + [_F, _P]),
    "msg_gather_taps": (_I, [_P, _P, _I] + [_I] * 9 + [_P]),
    "msg_fold_taps": (_I, [_P, _P, _P, _I] + [_I] * 10 + [_P]),
    "msg_scaled_add": (_I, [_P, _P, _P, _I, _L, _F, _F, _P]),
    "msg_rgb_skip_merge": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "msg_rgb_skip_merge_backward": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "msg_gamma_merge": (_I, [_P, _P, _P, _P, _I, _L, _F, _P]),
    "msg_gamma_merge_backward_workspace": (_L, []),
    "msg_gamma_merge_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _F, _P, _P]),
    "msg_scaled_add_rows": (_I, [_P, _P, _P, _I, _L, _I, _L, _L, _L, _F, _F, _P]),
    "msg_flat_adam": (_I, [_P, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _I, _F, _P]),
    "msg_flat_ema": (_I, [_P, _P, _L, _F, _P]),
    "msg_softmax_rows": (_I, [_P, _P, _I, _L, _I, _P]),
    "msg_softmax_rows_backward": (_I, [_P, _P, _P, _I, _L, _I, _P]),
    "msg_softmax_rows_backward2": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _P]),
    "msg_nonlocal_attention_supported": (_I, [_I] * 5),
    "msg_nonlocal_attention_fwd": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "msg_nonlocal_attention_bwd_splits": (_I, [_I] * 3),
    "msg_nonlocal_attention_bwd": (_I, [_P] * 14 + [_I] * 6 + [_P]),
    "msg_affine_warp": (_I, [_P, _P, _P, _F, _P, _P, _P, _I, _F, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "msg_minibatch_stddev_workspace": (_L, [_I] * 5),
    "msg_minibatch_stddev": (_I, [_P, _P, _P, _P, _I] + [_I] * 7 + [_F, _P]),
    "msg_minibatch_stddev_backward": (_I, [_P, _P, _P, _P, _I] + [_I] * 8 + [_F, _P]),
    "msg_conv2d_fprop_plan": (_I, [_I] * 11 + [_L]),
    "msg_conv2d_fprop_upconv_eligible": (_I, [_I] * 14 + [_L]),
    "msg_conv2d_fprop_thin_eligible": (_I, [_I] * 16),
    "msg_maxpool2x2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _L, _P]),
    "msg_maxpool2x2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "msg_maxpool2x2_gather": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _L, _P]),
    "msg_linear_fprop": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    "msg_linear_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "msg_linear_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    "msg_linear_grouped_fprop": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "msg_linear_grouped_dgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "msg_linear_grouped_wgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "msg_linear_grouped_wgrad_ptrs": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
}


def declared_symbols():
    """Every function the public header declares (used by the CPU-side export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msg_[a-z0-9_]+)\s*\(", text)))


class MsgHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MsgHipError(f"{LIB_PATH} is missing: build it with `python -m multi_stylegan_amd.build` "
                              "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.msg_abi_version.restype = _I
        found = handle.msg_abi_version()
        if found != ABI_VERSION:
            # (a stale .so used to surface as an AttributeError on the first missing symbol -- or not at all, when only a
            #  workspace size had changed)
            raise MsgHipError(f"{LIB_PATH} was built for C-ABI version {found}, this package binds version {ABI_VERSION}: "
                              "rebuild the library with `python -m multi_stylegan_amd.build --force`")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = _with_fastcall(handle)
    return _lib


class _Entries:
    """What ``lib()`` hands out: every entry point of the C ABI as an attribute -- the generated argument-vector wrapper
    (multi_stylegan_amd._msg_fastcall, csrc_host/gen_fastcall.py) where there is one, the ctypes function otherwise."""

    def __init__(self, handle, fast):
        self._ctypes, self.fastcall = handle, fast is not None
        for name in _SIGNATURES:
            setattr(self, name, getattr(fast, name, None) or getattr(handle, name))

    def __getattr__(self, name):                    # (anything else the library exports: resolved by ctypes on first use)
        return getattr(self._ctypes, name)


def _with_fastcall(handle):
    """ctypes boxes and converts every argument of a call (2.3-2.5 us per 24-argument launch); the generated wrappers read the
    Python ints / floats straight off the argument vector and call the SAME symbols of the SAME dlopen handle.  Optional:
    MSG_NO_FASTCALL=1 or a missing module (an interpreter the build did not see) leaves the ctypes binding in place."""
    if os.environ.get("MSG_NO_FASTCALL"):
        return _Entries(handle, None)
    try:
        from . import _msg_fastcall as fast
    except ImportError:
        return _Entries(handle, None)
    bound = fast.bind(handle._handle)
    want = sum(1 for res, args in _SIGNATURES.values() if res in (_I, _L))
    if bound != want:
        raise MsgHipError(f"_msg_fastcall resolved {bound} of {want} entry points of {LIB_PATH}: rebuild with "
                          "`python -m multi_stylegan_amd.build --force`")
    return _Entries(handle, fast)


def check(code, what):
    if code != 0:
        raise MsgHipError(f"{what}: {lib().msg_strerror(code).decode()} (code {code})")


def dtype_code(t: torch.Tensor, allow_half: bool = False, allow_double: bool = False) -> int:
    """MSG_* storage code of a tensor.  float16 only where `allow_half`, float64 only where `allow_double` (the FIR /
    activation entries that replace the reference's CUDA modules, which dispatch float / double / half:
    upfirdn2d_kernel.cu:225)."""
    if t.dtype == torch.float32:
        return MSG_F32
    if t.dtype == torch.bfloat16:
        return MSG_BF16
    if allow_half and t.dtype == torch.float16:
        return MSG_F16
    if allow_double and t.dtype == torch.float64:
        return MSG_F64
    raise MsgHipError(f"dtype {t.dtype} is not supported by the gfx950 kernels (float32 / bfloat16"
                      f"{' / float16' if allow_half else ''} only)")


def require_gpu(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MsgHipError("multi_stylegan_amd ops need tensors on an MI355X (ROCm 'cuda' device); "
                              "got a CPU tensor and there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise MsgHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_of(device) -> int:
    """Raw hipStream_t of torch's current stream on `device` (the fast accessor: no Stream object is built)."""
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """Context that makes `device` current for the launch.  One process drives one GPU, so the device is already
    current in practice and the guard costs one integer comparison instead of two hipSetDevice round trips."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


_SCRATCH: dict = {}


def scratch_ptr(nfloats: int, device) -> int:
    """Device pointer of at least `nfloats` fp32 words of launch-scoped workspace (per-tile partial sums, K-slice slabs): ONE
    buffer per (device, stream), grown on demand and reused by every launch.  Sound because a workspace lives from a
    kernel to its reduce launch, both enqueued by the same call on the same stream, and the next user is ordered behind them on
    that stream; a fresh torch.empty per launch was ~1.5 us of host time x ~300 launches per iteration.  (A buffer that was
    outgrown stays alive until the process ends: launches already enqueued may still be reading it.)"""
    if nfloats <= 0:
        return 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_of(device))
    bufs = _SCRATCH.get(key)
    if bufs is None or bufs[-1].numel() < nfloats:
        grown = torch.empty(max(int(nfloats), 1 << 20, 2 * (bufs[-1].numel() if bufs else 0)), dtype=torch.float32, device=device)
        bufs = (bufs or []) + [grown]
        _SCRATCH[key] = bufs
    return bufs[-1].data_ptr()


class KernelClock:
    """Optional per-launch timing with HIP events on the stream the kernels are launched on (bench.py's roofline
    leg).  Off by default: when off, ``span`` is a no-op and costs one attribute test."""

    def __init__(self):
        self.enabled = False
        self.only = None
        self.spans = {}

    def reset(self, enabled: bool, only=None):
        """only: tuple of key prefixes to time (None = every launch; two events per timed launch cost host time)."""
        self.enabled, self.spans, self.only = enabled, {}, (tuple(only) if only else None)

    def span(self, key, work: float):
        """``key``: the timing label, or a tuple of its '/'-separated parts -- joined only when the clock is on, so that a
        disabled clock costs the launch sites one attribute test and no string formatting."""
        if not self.enabled:
            return _NULL_SPAN
        if not isinstance(key, str):
            key = "/".join("".join(map(str, part)) if isinstance(part, tuple) else str(part) for part in key)
        if self.only is not None and not key.startswith(self.only):
            return _NULL_SPAN
        return _Span(self, key, work)

    def summary(self):
        """-> {key: dict(launches, total_ms, avg_us, work)}; call after torch.cuda.synchronize()."""
        out = {}
        for key, items in self.spans.items():
            total_ms = sum(a.elapsed_time(b) for a, b, _ in items)
            work = sum(w for _, _, w in items)
            out[key] = {"launches": len(items), "total_ms": total_ms, "avg_us": 1e3 * total_ms / len(items),
                        "work": work}
        return out


class _Span:
    __slots__ = ("clock", "key", "work", "start")

    def __init__(self, clock, key, work):
        self.clock, self.key, self.work = clock, key, work

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def __exit__(self, *exc):
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.clock.spans.setdefault(self.key, []).append((self.start, end, self.work))
        return False


class _NullSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_SPAN = _NullSpan()
kernel_clock = KernelClock()
