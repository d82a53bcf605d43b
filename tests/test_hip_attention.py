"""Fused non-local attention (csrc/attention.hip) against the CPU oracle's restatement of
multi_stylegan/u_net_2d_discriminator.py:376-380: forward, first-order gradients (fused kernels), second-order
gradients (composite path), at the sizes of the 256^2 (4096 x 1024) and 512^2 (16384 x 4096) configurations."""
import math

import pytest
import torch

from conftest import rel_err
from oracle import models as om
from oracle import ops as oo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# relative to max|ref|: fp32 storage runs exact-fp32 MFMA; bf16 storage rounds q, k, v, P and dS to 8 bits
TOL = {torch.float32: (2e-5, 1e-4), torch.bfloat16: (2e-2, 4e-2)}       # (forward, gradients)


def _oracle(q, k, v):
    """oracle.ops.non_local_attention takes the reference's channel-first operands (theta [B,c8,HW], phi [B,c8,HW/4],
    g [B,c2,HW/4]) and returns [B,c2,HW]."""
    return oo.non_local_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)


def _inputs(b, nq, nk, dk, dv, dtype, seed, spike=False):
    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(b, nq, dk, generator=gen) * 0.5
    k = torch.randn(b, nk, dk, generator=gen) * 0.5
    v = torch.randn(b, nk, dv, generator=gen)
    if spike:           # rows whose maximum is far above the rest, and a key that dominates every row of one sample
        q[:, ::7] *= 12.0
        k[0, 5] *= 9.0
    q, k, v = (t.to(dtype).float() for t in (q, k, v))                    # the values the storage type can hold
    return q, k, v


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 4096, 1024, 48, 192), (1, 16384, 4096, 48, 192), (3, 256, 128, 16, 64),
                                   (2, 512, 384, 48, 192)])
@pytest.mark.parametrize("spike", [False, True])
def test_fused_attention_forward_backward(shape, dtype, spike):
    from multi_stylegan_amd.op_static import attention, non_local_attention
    b, nq, nk, dk, dv = shape
    if spike and nq > 4096:
        pytest.skip("spiked rows are covered at the smaller sizes")
    q, k, v = _inputs(b, nq, nk, dk, dv, dtype, seed=nq + dk, spike=spike)
    go = torch.randn(b, nq, dv, generator=torch.Generator().manual_seed(1)).to(dtype).float()
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    want = _oracle(qc, kc, vc)
    want.backward(go)
    qd, kd, vd = (t.to(DEV, dtype).requires_grad_(True) for t in (q, k, v))
    assert attention.supported(qd, kd, vd)
    got = non_local_attention(qd, kd, vd)
    assert got.dtype == dtype and got.shape == (b, nq, dv)
    got.backward(go.to(DEV, dtype))
    tf, tg = TOL[dtype]
    assert rel_err(got, want) < tf
    for name, a, r in (("dq", qd.grad, qc.grad), ("dk", kd.grad, kc.grad), ("dv", vd.grad, vc.grad)):
        assert torch.isfinite(a).all(), name
        assert rel_err(a, r) < tg, (name, rel_err(a, r))
    # deterministic (no atomics): a second run gives the same bits
    q2, k2, v2 = (t.detach().clone().requires_grad_(True) for t in (qd, kd, vd))
    again = non_local_attention(q2, k2, v2)
    again.backward(go.to(DEV, dtype))
    assert torch.equal(again, got) and torch.equal(q2.grad, qd.grad) and torch.equal(k2.grad, kd.grad) \
        and torch.equal(v2.grad, vd.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_attention_second_order(dtype):
    """R1-style double backward through the fused op: d/d(k, v) of |d(sum(o * w)) / dq|^2 (fused forward, composite
    gradient graph) against the oracle's autograd."""
    from multi_stylegan_amd.op_static import non_local_attention
    b, nq, nk, dk, dv = 2, 256, 128, 48, 192
    q, k, v = _inputs(b, nq, nk, dk, dv, dtype, seed=3)
    w = torch.randn(b, nq, dv, generator=torch.Generator().manual_seed(2)).to(dtype).float()

    def penalty(fn, q_, k_, v_, w_):
        gq, = torch.autograd.grad((fn(q_, k_, v_).float() * w_.float()).sum(), q_, create_graph=True)
        return gq.float().square().sum()
    qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
    want = torch.autograd.grad(penalty(_oracle, qc, kc, vc, w), (qc, kc, vc))
    qd, kd, vd = (t.to(DEV, dtype).requires_grad_(True) for t in (q, k, v))
    got = torch.autograd.grad(penalty(non_local_attention, qd, kd, vd, w.to(DEV)), (qd, kd, vd))
    tol = 1e-3 if dtype == torch.float32 else 6e-2
    for a, r in zip(got, want):
        assert rel_err(a, r) < tol, rel_err(a, r)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_non_local_block_uses_fused_attention(dtype):
    """The product NonLocalBlock at the reference's channel counts (256 -> 384: dk 48, dv 192) on a 64 x 64 map runs the
    fused kernels and matches the oracle's block, forward and backward, with a non-zero gamma."""
    from multi_stylegan_amd import conv_ops, u_net_2d_discriminator as U
    from multi_stylegan_amd.op_static import attention
    torch.manual_seed(11)
    ref = om.NonLocalBlock(256, 384)
    with torch.no_grad():
        ref.gamma.fill_(0.7)
    blk = U.NonLocalBlock(256, 384)
    blk.load_state_dict(ref.state_dict())
    blk.to(DEV)
    x = torch.randn(2, 256, 64, 64) * 0.5
    gy = torch.randn(2, 384, 64, 64)
    xr = x.clone().requires_grad_(True)
    want = ref(xr)
    want.backward(gy)
    calls = []
    orig = attention._NonLocalAttention.apply
    attention._NonLocalAttention.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        xd = conv_ops.to_compute_layout(x.to(DEV), dtype).requires_grad_(True)
        got = blk(xd)
        got.backward(conv_ops.to_compute_layout(gy.to(DEV), dtype))
    finally:
        attention._NonLocalAttention.apply = orig
    assert calls, "the fused attention path did not run"
    tol = 1e-3 if dtype == torch.float32 else 4e-2
    assert rel_err(got.float(), want) < tol
    if dtype == torch.float32:
        assert rel_err(xd.grad.float(), xr.grad) < tol
    else:
        # bf16 rounds many 2x2 max-pool windows into ties: the gradient of a tied window goes to another pixel than in
        # the fp32 oracle, an element-wise difference as large as the element itself -- compare in the norm
        err = ((xd.grad.float().cpu() - xr.grad).norm() / xr.grad.norm()).item()
        assert err < 5e-2, err
    for (n, p), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        if dtype == torch.float32:
            assert rel_err(p.grad, pr.grad) < 2e-3, n
        else:       # the oracle's theta / phi / g maps are not rounded to bf16 before the attention: norm-wise
            err = ((p.grad.cpu() - pr.grad).norm() / pr.grad.norm()).item()
            assert err < 6e-2, (n, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,groups", [((16, 768, 32, 32), 1), ((32, 1024, 16, 16), 2), ((6, 24, 5, 7), 3),
                                           ((4, 32, 8, 8), 1)])
def test_fused_minibatch_stddev(shape, groups, dtype):
    """csrc/mbstd.hip against the oracle (u_net_2d_discriminator.py:205-217), per group: forward (statistic plane +
    the concatenation), first-order gradient (fused kernel) and an R1-style second-order gradient (composite path);
    run twice: bit-identical (fixed-order sums)."""
    from multi_stylegan_amd import conv_ops, u_net_2d_discriminator as U
    torch.manual_seed(shape[1])
    x = (torch.randn(shape) * torch.rand(1, shape[1], 1, 1) * 2).to(dtype).float()
    x[:, 0] = 0.25                                       # a channel without variance: the clamp, zero gradient
    n = shape[0] // groups
    xr = x.clone().requires_grad_(True)
    want = torch.cat([oo.minibatch_stddev(xr[g * n:(g + 1) * n]) for g in range(groups)])
    gy = torch.randn(want.shape).to(dtype).float()
    want.backward(gy)
    mod = U.MinibatchStdDev()
    mod.groups = groups
    xd = conv_ops.to_compute_layout(x.to(DEV), dtype).requires_grad_(True)
    got = mod(xd)
    assert got.shape == want.shape and got.dtype == dtype
    got.backward(conv_ops.to_compute_layout(gy.to(DEV), dtype))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(got, want) < tol and rel_err(got[:, -1], want[:, -1]) < tol
    assert rel_err(xd.grad, xr.grad) < tol
    x2 = xd.detach().clone().requires_grad_(True)
    again = mod(x2)
    again.backward(conv_ops.to_compute_layout(gy.to(DEV), dtype))
    assert torch.equal(again, got) and torch.equal(x2.grad, xd.grad)
    if shape[2] <= 8:           # second order: d/dx of |d(sum(y * w))/dx|^2
        w = torch.randn(want.shape).to(dtype).float()

        def pen(fn, x_):        # (y^2 w: the statistic plane enters non-linearly, its second derivative matters)
            g, = torch.autograd.grad((fn(x_).float().square() * w.to(x_.device)).sum(), x_, create_graph=True)
            return g.float().square().sum()
        x3 = x.clone().requires_grad_(True)
        ref2, = torch.autograd.grad(pen(lambda t: torch.cat([oo.minibatch_stddev(t[g * n:(g + 1) * n])
                                                             for g in range(groups)]), x3), x3)
        x4 = conv_ops.to_compute_layout(x.to(DEV), dtype).requires_grad_(True)
        got2, = torch.autograd.grad(pen(mod, x4), x4)
        assert rel_err(got2, ref2) < (1e-3 if dtype == torch.float32 else 5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_attention_shape_sweep(dtype):
    """Every combination of query / key counts that changes the launch structure (one or several key blocks, query
    sweep split 1 / 2 / 4 ways, odd batch sizes), both head sizes, against the oracle."""
    from multi_stylegan_amd import _lib
    from multi_stylegan_amd.op_static import attention, non_local_attention
    tf, tg = TOL[dtype]
    seen_splits = set()
    for (b, nq, nk, dk, dv) in ((1, 128, 128, 48, 192), (5, 1024, 128, 48, 192), (3, 2048, 256, 16, 64),
                                (2, 4096, 128, 48, 192), (7, 640, 384, 16, 64), (1, 8192, 256, 48, 192)):
        seen_splits.add(_lib.lib().msg_nonlocal_attention_bwd_splits(b, nq, nk))
        q, k, v = _inputs(b, nq, nk, dk, dv, dtype, seed=b * nq + nk)
        go = torch.randn(b, nq, dv, generator=torch.Generator().manual_seed(b)).to(dtype).float()
        qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
        want = _oracle(qc, kc, vc)
        want.backward(go)
        qd, kd, vd = (t.to(DEV, dtype).requires_grad_(True) for t in (q, k, v))
        assert attention.supported(qd, kd, vd)
        got = non_local_attention(qd, kd, vd)
        got.backward(go.to(DEV, dtype))
        assert rel_err(got, want) < tf, (b, nq, nk)
        for a, r in ((qd.grad, qc.grad), (kd.grad, kc.grad), (vd.grad, vc.grad)):
            assert rel_err(a, r) < tg, (b, nq, nk, rel_err(a, r))
    assert {1, 2, 4} <= seen_splits or {1, 2, 8} <= seen_splits or len(seen_splits) >= 3, seen_splits
    # shapes the kernels do not take fall back to the composite path, silently and correctly
    q, k, v = _inputs(2, 100, 60, 24, 40, dtype, seed=1)
    qd, kd, vd = (t.to(DEV, dtype) for t in (q, k, v))
    assert not attention.supported(qd, kd, vd)
    assert rel_err(non_local_attention(qd, kd, vd), _oracle(q, k, v)) < (1e-4 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 48, 64, 64), (2, 192, 16, 24), (1, 8, 2, 2)])
def test_max_pool2x2_matches_library(shape, dtype):
    """csrc/maxpool.hip against F.max_pool2d (u_net_2d_discriminator.py:366-370): output and gradient bit for bit -- including
    windows with TIED maxima, where the gradient must go to the first maximum in scan order as the library sends it --
    on dense maps and on a channel-slice, and through a second-order pass."""
    import torch.nn.functional as F
    from multi_stylegan_amd.op_static import max_pool2x2
    torch.manual_seed(shape[1])
    x = (torch.randn(*shape, device=DEV) * 2).round() / 2          # coarse values: many ties
    x = x.to(dtype).contiguous(memory_format=torch.channels_last)
    for src in (x, torch.cat([x, x], dim=1).contiguous(memory_format=torch.channels_last)[:, shape[1]:]):
        a = src.clone().requires_grad_(True) if src is x else src.detach().requires_grad_(True)
        b = src.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ya, yb = max_pool2x2(a), F.max_pool2d(b, kernel_size=2, stride=2)
        assert torch.equal(ya, yb)
        gy = torch.randn_like(yb)
        ga, = torch.autograd.grad(ya, a, gy)
        gb, = torch.autograd.grad(yb, b, gy)
        assert torch.equal(ga, gb)
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    gy = torch.randn(shape[0], shape[1], shape[2] // 2, shape[3] // 2, device=DEV, dtype=dtype).requires_grad_(True)
    for t, pool in ((a, max_pool2x2), (b, lambda v: F.max_pool2d(v, kernel_size=2, stride=2))):
        g1, = torch.autograd.grad(pool(t), t, gy, create_graph=True)
        t.second = torch.autograd.grad(g1.square().sum(), gy)[0]
    assert torch.equal(a.second, b.second)


def test_max_pool2x2_declined_shapes_take_the_library():
    """Shapes the pooling kernel declines -- odd height / width, a channel count that is not a whole number of 16-byte vectors,
    a map that is not channels-last -- go to F.max_pool2d on the same device: same values, differentiable (review, round 4:
    the branch existed untested)."""
    import torch.nn.functional as F
    from multi_stylegan_amd.op_static import max_pool2x2
    torch.manual_seed(3)
    for shape, cl in (((2, 8, 5, 6), True), ((2, 8, 6, 7), True), ((1, 6, 4, 4), True), ((2, 16, 8, 8), False)):
        x = torch.randn(*shape, device=DEV)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = max_pool2x2(a), F.max_pool2d(b, kernel_size=2, stride=2)
        assert ya.shape == yb.shape and torch.equal(ya, yb), shape
        gy = torch.randn_like(yb)
        assert torch.equal(torch.autograd.grad(ya, a, gy)[0], torch.autograd.grad(yb, b, gy)[0]), shape
