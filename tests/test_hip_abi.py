"""Error behaviour of the C ABI (include/msg_hip.h), called directly through ctypes on the GPU box: every entry point
rejects null / negative / misaligned / unsupported arguments with a status code instead of launching, and accepts an
empty batch as a no-op.  (The reference's native ops TORCH_CHECK their inputs, upfirdn2d.cpp:15-16,
fused_bias_act.cpp:13-14, and silently launch nothing for unsupported modes -- SURVEY Q14.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OK, EINVAL, EUNSUPPORTED = 0, -1, -2
F32, BF16, F64 = 0, 1, 7


@pytest.fixture(scope="module")
def lib():
    from multi_stylegan_amd import _lib
    return _lib.lib()


def _buf(n=4096, dtype=torch.float32):
    return torch.zeros(n, device=DEV, dtype=dtype)


def test_version_and_strerror(lib):
    assert lib.msg_abi_version() >= 2 and lib.msg_build_arch() == b"gfx950"
    for code in (OK, EINVAL, EUNSUPPORTED, -3, 12345):
        assert isinstance(lib.msg_strerror(code), bytes) and len(lib.msg_strerror(code)) > 0


def test_upfirdn2d_entries(lib):
    x, y, fir = _buf(), _buf(), _buf(16)
    s = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), fir.data_ptr(), y.data_ptr(), F32, 1, 8, 8, 4, 4, 4, 1, 1, 1, 1, 2, 1, 2, 1, s)
    assert lib.msg_upfirdn2d(*args) == OK
    assert lib.msg_upfirdn2d(None, *args[1:]) == EINVAL                              # null input
    assert lib.msg_upfirdn2d(*args[:3], F64, *args[4:]) == EUNSUPPORTED              # storage type
    assert lib.msg_upfirdn2d(*args[:10], 0, *args[11:]) == EINVAL                    # up_x = 0
    assert lib.msg_upfirdn2d(*args[:4], 0, *args[5:]) == OK                          # empty batch: no-op
    assert lib.msg_upfirdn2d(x.data_ptr(), fir.data_ptr(), y.data_ptr(), F32, 1, 2, 2, 4, 4, 4, 1, 1, 1, 1,
                             0, 0, 0, 0, s) == EINVAL                                # empty output
    # pitched: pitch below the channel count is invalid
    assert lib.msg_upfirdn2d_pitched(x.data_ptr(), fir.data_ptr(), y.data_ptr(), F32, 1, 8, 8, 4, 2, 4, 4, 1, 1, 1, 1,
                                     2, 1, 2, 1, s) == EINVAL
    # separable: only 4x4, whole vectors, aligned
    fy = _buf(4)
    sep = (x.data_ptr(), fy.data_ptr(), fy.data_ptr(), y.data_ptr(), F32, 1, 8, 8, 4, 4, 4, 2, 1, 2, 1, s)
    assert lib.msg_upfirdn2d_separable(*sep) == OK
    assert lib.msg_upfirdn2d_separable(*sep[:9], 3, 3, *sep[11:]) == EUNSUPPORTED    # 3x3 FIR
    assert lib.msg_upfirdn2d_separable(*sep[:8], 3, *sep[9:]) == EUNSUPPORTED        # 3 channels: not a whole vector
    assert lib.msg_upfirdn2d_separable(x.data_ptr() + 4, *sep[1:]) == EUNSUPPORTED   # misaligned
    assert lib.msg_upfirdn2d_separable(*sep[:5], 0, *sep[6:]) == OK                  # empty batch


def test_bias_act_entries(lib):
    x, y, b = _buf(), _buf(), _buf(8)
    s = torch.cuda.current_stream().cuda_stream
    ok = (x.data_ptr(), b.data_ptr(), None, y.data_ptr(), F32, 64, 8, 8, None, None, 1, 1, 3, 0, 0.2, 1.0, s)
    assert lib.msg_fused_bias_act(*ok) == OK
    assert lib.msg_fused_bias_act(*ok[:12], 2, *ok[13:]) == EINVAL                   # act code the reference lacks
    assert lib.msg_fused_bias_act(*ok[:13], 1, *ok[14:]) == EINVAL                   # grad = 1 needs the reference map
    assert lib.msg_fused_bias_act(*ok[:4], F64, *ok[5:]) == EUNSUPPORTED
    assert lib.msg_fused_bias_act(*ok[:5], 0, *ok[6:]) == OK                         # empty
    assert lib.msg_scaled_add(x.data_ptr(), y.data_ptr(), y.data_ptr(), F32, 6, 1.0, 1.0, s) == EUNSUPPORTED   # ragged
    assert lib.msg_scaled_add(None, y.data_ptr(), y.data_ptr(), F32, 8, 1.0, 1.0, s) == EINVAL
    assert lib.msg_scaled_add_rows(x.data_ptr(), y.data_ptr(), y.data_ptr(), F32, 4, 8, 4, 8, 8, 1.0, 1.0, s) == EINVAL


def test_conv_entries(lib):
    x, w, y = _buf(1 << 16, torch.bfloat16), _buf(1 << 16, torch.bfloat16), _buf(1 << 16, torch.bfloat16)
    gw = _buf(1 << 16)
    s = torch.cuda.current_stream().cuda_stream
    ok = (x.data_ptr(), w.data_ptr(), None, y.data_ptr(), BF16, 1, 8, 8, 64, 64, 8, 8, 16, 16, 3, 3, 1, 1, 1, 0, 0, s)
    assert lib.msg_conv2d_fprop(*ok) == OK
    assert lib.msg_conv2d_fprop(*ok[:9], 48, *ok[10:]) == EUNSUPPORTED               # K extent not a 128-byte run
    assert lib.msg_conv2d_fprop(*ok[:8], 3, *ok[9:]) == EUNSUPPORTED                 # channel pitch not whole vectors
    assert lib.msg_conv2d_fprop(*ok[:4], F64, *ok[5:]) == EUNSUPPORTED
    assert lib.msg_conv2d_fprop(*ok[:5], 0, *ok[6:]) == OK                           # empty batch
    assert lib.msg_conv2d_fprop(None, *ok[1:]) == EINVAL
    assert lib.msg_conv2d_fprop(*ok[:16], 2, ok[17], 2, *ok[19:]) == EUNSUPPORTED    # zero-insertion AND a stride
    # fused activation: noise without its weight is invalid
    act = (x.data_ptr(), w.data_ptr(), y.data_ptr(), BF16, 1, 8, 8, 64, 64, 8, 8, 16, 16, 3, 3, 1, 1, 0)
    assert lib.msg_conv2d_fprop_act(*act, None, gw.data_ptr(), None, 1, 0.2, 1.0, s) == EINVAL
    assert lib.msg_conv2d_fprop_act(*act, None, None, None, 1, 0.2, 1.0, s) == OK
    assert lib.msg_conv2d_fprop_residual(*act, None, 16, 1.0, s) == EINVAL           # no residual map
    assert lib.msg_conv2d_fprop_residual(*act, y.data_ptr(), 8, 1.0, s) == EINVAL    # residual pitch below N
    wg = (y.data_ptr(), x.data_ptr(), gw.data_ptr(), BF16, 1, 8, 8, 64, 64, 8, 8, 16, 16, 64, 3, 3, 1, 1, 0, 0, 1, 0, 1.0, None, 0, s)
    assert lib.msg_conv2d_wgrad(*wg) == OK
    assert lib.msg_conv2d_wgrad(*wg[:13], 32, *wg[14:]) == EINVAL                    # gradient pitch below I
    assert lib.msg_conv2d_wgrad(*wg[:20], 0, *wg[21:]) == EINVAL                     # zero K chunks
    torch.cuda.synchronize()


def test_linear_and_modulation_entries(lib):
    a, b, c, d = _buf(), _buf(), _buf(), _buf()
    s = torch.cuda.current_stream().cuda_stream
    assert lib.msg_linear_fprop(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), 4, 8, 16, 1.0, 1.0, s) == OK
    assert lib.msg_linear_fprop(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), 4, 0, 16, 1.0, 1.0, s) == EINVAL
    assert lib.msg_linear_fprop(None, None, None, None, 0, 8, 16, 1.0, 1.0, s) == OK                     # empty batch
    assert lib.msg_linear_dgrad(None, b.data_ptr(), c.data_ptr(), 4, 8, 16, 1.0, s) == EINVAL
    assert lib.msg_linear_wgrad(None, None, c.data_ptr(), None, 0, 8, 16, 1.0, 1.0, s) == OK             # zero gradient
    torch.cuda.synchronize()
    assert float(c[:128].abs().max()) == 0.0
    assert lib.msg_relayout_weight(a.data_ptr(), b.data_ptr(), None, None, F32, 8, 4, 25, 8, 8, 1, 0, 1.0, s) == EUNSUPPORTED
    assert lib.msg_relayout_weight(a.data_ptr(), b.data_ptr(), None, None, F32, 8, 8, 9, 4, 8, 1, 0, 1.0, s) == EINVAL
    assert lib.msg_modulate_weights(a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), None, F32, 2, 8, 3, 1, 8, 8,
                                    1.0, 1e-8, s) == EINVAL                                               # R % O != 0
    assert lib.msg_scale_rows_cols(a.data_ptr(), None, None, d.data_ptr(), F32, 2, 8, 1, 8, 4, 1.0, s) == EINVAL  # Ck < C
    torch.cuda.synchronize()


def test_softmax_and_grouped_linear_entries(lib):
    a, b, c = _buf(1 << 16), _buf(1 << 16), _buf(1 << 16)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.msg_softmax_rows(a.data_ptr(), b.data_ptr(), F32, 4, 1024, s) == OK
    assert lib.msg_softmax_rows(a.data_ptr(), b.data_ptr(), F32, 0, 1024, s) == OK                  # no rows: no-op
    assert lib.msg_softmax_rows(a.data_ptr(), b.data_ptr(), F32, 4, 8192, s) == EUNSUPPORTED        # row > one wave's registers
    assert lib.msg_softmax_rows(a.data_ptr(), b.data_ptr(), F32, 4, 1022, s) == EUNSUPPORTED        # not whole 16-byte vectors
    assert lib.msg_softmax_rows(a.data_ptr(), b.data_ptr(), F64, 4, 1024, s) == EUNSUPPORTED
    assert lib.msg_softmax_rows(None, b.data_ptr(), F32, 4, 1024, s) == EINVAL
    assert lib.msg_softmax_rows(a.data_ptr() + 4, b.data_ptr(), F32, 4, 1024, s) == EUNSUPPORTED    # misaligned
    assert lib.msg_softmax_rows_backward(a.data_ptr(), b.data_ptr(), None, F32, 4, 1024, s) == EINVAL
    assert lib.msg_softmax_rows_backward(a.data_ptr(), b.data_ptr(), c.data_ptr(), BF16, 4, 4096, s) == OK
    torch.cuda.synchronize()
    assert abs(float(b[:1024].sum()) - 1.0) < 1e-5                                                   # softmax of zeros
    # grouped linear: pointer tables on the device
    w = [_buf(64) for _ in range(3)]
    wt = torch.tensor([t.data_ptr() for t in w], dtype=torch.int64, device=DEV)
    slot = torch.tensor([0, 1, 1], dtype=torch.int32, device=DEV)
    x, y = _buf(2 * 2 * 8), _buf(3 * 2 * 8)
    args = (x.data_ptr(), slot.data_ptr(), wt.data_ptr(), None, y.data_ptr())
    assert lib.msg_linear_grouped_fprop(*args, 3, 2, 8, 8, 2, 1.0, 1.0, s) == OK
    assert lib.msg_linear_grouped_fprop(*args, 0, 2, 8, 8, 2, 1.0, 1.0, s) == OK                     # no groups: no-op
    assert lib.msg_linear_grouped_fprop(*args, 3, 2, 8, 8, 0, 1.0, 1.0, s) == EINVAL                 # latent slots per row
    assert lib.msg_linear_grouped_fprop(x.data_ptr(), None, wt.data_ptr(), None, y.data_ptr(), 3, 2, 8, 8, 2, 1.0, 1.0,
                                        s) == EINVAL
    assert lib.msg_linear_grouped_dgrad(y.data_ptr(), None, x.data_ptr(), 3, 2, 8, 8, 1.0, s) == EINVAL
    assert lib.msg_linear_grouped_wgrad(y.data_ptr(), x.data_ptr(), slot.data_ptr(), None, None, 3, 2, 8, 8, 2, 1.0, 1.0,
                                        s) == EINVAL
    torch.cuda.synchronize()


def test_round3_entries(lib):
    """The entries added in round 3 reject what they cannot take instead of launching."""
    s = torch.cuda.current_stream().cuda_stream
    x, y, fir = _buf(), _buf(), _buf(16)
    # output pitch below the channel count
    assert lib.msg_upfirdn2d_pitched2(x.data_ptr(), fir.data_ptr(), y.data_ptr(), F32, 1, 8, 8, 4, 4, 2, 4, 4, 1, 1, 1, 1,
                                      2, 1, 2, 1, s) == EINVAL
    assert lib.msg_upfirdn2d_pitched2(x.data_ptr(), fir.data_ptr(), y.data_ptr(), F32, 1, 8, 8, 4, 4, 8, 4, 4, 1, 1, 1, 1,
                                      2, 1, 2, 1, s) == OK
    # channel sums: size not a multiple of C / missing workspace / ragged channel vectors
    ws = _buf(1 << 16)
    assert lib.msg_channel_sums(x.data_ptr(), y.data_ptr(), F32, 4096, 8, ws.data_ptr(), ws.numel(), s) == OK
    assert lib.msg_channel_sums(x.data_ptr(), y.data_ptr(), F32, 4095, 8, ws.data_ptr(), ws.numel(), s) == EINVAL
    assert lib.msg_channel_sums(x.data_ptr(), y.data_ptr(), F32, 4096, 8, None, 0, s) == EINVAL
    assert lib.msg_channel_sums(x.data_ptr(), y.data_ptr(), F32, 4092, 3, ws.data_ptr(), ws.numel(), s) == EUNSUPPORTED
    # max-pooling: odd maps, missing index map in backward
    idx = torch.zeros(4096, device=DEV, dtype=torch.int16)
    assert lib.msg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), F32, 1, 8, 8, 8, 8, s) == OK
    assert lib.msg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), F32, 1, 7, 8, 8, 8, s) == EUNSUPPORTED
    assert lib.msg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), F32, 1, 8, 8, 8, 4, s) == EINVAL      # pitch < C
    assert lib.msg_maxpool2x2_bwd(x.data_ptr(), None, y.data_ptr(), F32, 1, 8, 8, 8, s) == EINVAL
    assert lib.msg_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), None, F32, 0, 8, 8, 8, 8, s) == OK                     # empty batch
    # gamma merge: null scalar, ragged length
    g = _buf(1)
    assert lib.msg_gamma_merge(x.data_ptr(), y.data_ptr(), g.data_ptr(), y.data_ptr(), F32, 64, 0.5, s) == OK
    assert lib.msg_gamma_merge(x.data_ptr(), y.data_ptr(), None, y.data_ptr(), F32, 64, 0.5, s) == EINVAL
    assert lib.msg_gamma_merge(x.data_ptr(), y.data_ptr(), g.data_ptr(), y.data_ptr(), F32, 63, 0.5, s) == EUNSUPPORTED
    assert lib.msg_gamma_merge_backward(x.data_ptr(), y.data_ptr(), g.data_ptr(), y.data_ptr(), y.data_ptr(), g.data_ptr(), F32,
                                        64, 0.5, None, s) == EINVAL                                                 # no workspace
    # sign-byte activation backward: bf16 only, whole tiles
    m = torch.zeros(4096, device=DEV, dtype=torch.uint8)
    xb = _buf(4096, torch.bfloat16)
    args = (xb.data_ptr(), m.data_ptr(), 1, 64, xb.data_ptr(), BF16, 4096, 64, None, None, None, 1, 64, 0.2, 1.0, None, 0, s)
    assert lib.msg_bias_act_backward_mask(*args) == OK
    assert lib.msg_bias_act_backward_mask(*args[:5], F32, *args[6:]) == EUNSUPPORTED
    assert lib.msg_bias_act_backward_mask(*args[:2], 3, 64, *args[4:]) == EINVAL          # 64 pixels are not whole tiles of 3
    assert lib.msg_bias_act_backward_mask(args[0], None, *args[2:]) == EINVAL
    # the masked conv entry refuses problems whose kernel does not write the bytes (ask msg_conv2d_fprop_plan first)
    w = _buf(64 * 64, torch.bfloat16)
    assert lib.msg_conv2d_fprop_act_mask(xb.data_ptr(), w.data_ptr(), xb.data_ptr(), BF16, 1, 8, 8, 64, 64, 8, 8, 64, 64, 1, 1, 1, 0,
                                         0, None, None, None, 1, 0.2, 1.0, m.data_ptr(), s) == EUNSUPPORTED
    torch.cuda.synchronize()
