"""Fused attention of the discriminator's NonLocalBlock on the gfx950 kernels (csrc/attention.hip), twice
differentiable.

Reference call site: multi_stylegan/u_net_2d_discriminator.py:376-380 (``beta = softmax(bmm(theta^T, phi))``,
``o = bmm(g, beta^T)``).  The [B, Nq, Nk] attention map is never written to HBM: forward keeps the per-row
log-sum-exp, the first-order backward recomputes the probabilities tile by tile.  A SECOND-order graph through the
block (R1 on the discriminator, every 16th iteration) is built from the composite formulation instead -- plain torch
products and softmax, recomputed inside backward -- which is the only place the map still materialises.
"""

import torch
from torch.autograd import Function

from .. import _lib

def supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    if not (q.is_cuda and q.dtype in (torch.float32, torch.bfloat16)
            and q.dtype == k.dtype == v.dtype and q.ndim == 3):
        return False
    return bool(_lib.lib().msg_nonlocal_attention_supported(q.shape[0], q.shape[1], k.shape[1], q.shape[2], v.shape[2]))


def composite(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T) v with library products around the row-softmax kernel (fp32 statistics, storage-type map; any
    shape, differentiable to any order).  This is where the [B, Nq, Nk] map still exists: shapes the fused kernels do
    not take, and the second-order graph of the R1 iteration."""
    from .softmax import softmax_rows
    return torch.bmm(softmax_rows(torch.bmm(q, k.transpose(1, 2))), v)


class _NonLocalAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v):          # dense inputs (non_local_attention makes them so, OUTSIDE, on the graph)
        dev = _lib.require_gpu(q, k, v)
        b, nq, dk = q.shape
        nk, dv = k.shape[1], v.shape[2]
        vt = v.transpose(1, 2).contiguous()
        o = torch.empty((b, nq, dv), dtype=q.dtype, device=dev)
        lse = torch.empty((b, nq), dtype=torch.float32, device=dev)
        flops = 2.0 * b * nq * nk * (2 * dk + dv)          # two score sweeps + P V
        with _lib.on_device(dev), _lib.kernel_clock.span(('nl_attention_fwd', q.dtype), flops):
            code = _lib.lib().msg_nonlocal_attention_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(),
                                                         lse.data_ptr(), _lib.dtype_code(q), b, nq, nk, dk, dv,
                                                         _lib.stream_of(dev))
        _lib.check(code, "msg_nonlocal_attention_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, grad_o):
        q, k, v, o, lse = ctx.saved_tensors
        if torch.is_grad_enabled():
            # a second-order graph is being built (create_graph=True): gradients as differentiable functions of
            # (q, k, v, grad_o) from the composite formulation
            with torch.enable_grad():       # the saved inputs carry their history: the result stays on the graph
                qq, kk, vv = (t if t.requires_grad else t.detach().requires_grad_(True) for t in (q, k, v))
                return torch.autograd.grad(composite(qq, kk, vv), (qq, kk, vv), grad_o, create_graph=True)
        grad_o = grad_o.contiguous()
        dev = q.device
        b, nq, dk = q.shape
        nk, dv = k.shape[1], v.shape[2]
        delta = torch.empty((b, nq), dtype=torch.float32, device=dev)        # filled by the dQ kernel
        kt = k.transpose(1, 2).contiguous()      # (Q^T and dO^T -- the big ones -- are read transposed inside the kernel)
        dq, dkey, dval = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        splits = _lib.lib().msg_nonlocal_attention_bwd_splits(b, nq, nk)
        work = torch.empty(splits * b * nk * (dk + dv), dtype=torch.float32, device=dev) if splits > 1 else None
        flops = 2.0 * b * nq * nk * ((dk + dv + dk) + (dk + dv + dv + dk))
        with _lib.on_device(dev), _lib.kernel_clock.span(('nl_attention_bwd', q.dtype), flops):
            code = _lib.lib().msg_nonlocal_attention_bwd(
                q.data_ptr(), None, k.data_ptr(), kt.data_ptr(), v.data_ptr(), grad_o.data_ptr(),
                None, o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dkey.data_ptr(),
                dval.data_ptr(),
                _lib.ptr(work), _lib.dtype_code(q), b, nq, nk, dk, dv, _lib.stream_of(dev))
        _lib.check(code, "msg_nonlocal_attention_bwd")
        return dq, dkey, dval


def non_local_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """o [B, Nq, dv] = softmax(q k^T) v for q [B, Nq, dk], k [B, Nk, dk], v [B, Nk, dv].  Shapes the fused kernels do
    not take (see include/msg_hip.h) run the composite formulation on the same device."""
    if supported(q, k, v):
        return _NonLocalAttention.apply(q.contiguous(), k.contiguous(), v.contiguous())
    return composite(q, k, v)
