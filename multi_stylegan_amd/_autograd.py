"""Calling this package's derivative ops from inside a first-order backward without the autograd machinery."""
import torch


class _NoCtx:
    """Stand-in for the autograd context when a derivative op of this package runs INSIDE a first-order backward (grad mode
    off): nothing is saved and no node is recorded, so the op's ``forward`` can be called as a plain function."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def _derive(fn, *args):
    """``fn.apply(*args)`` where a higher-order graph is being recorded, ``fn.forward`` called directly otherwise: inside a
    first-order backward ``Function.apply`` still builds a context, wraps and unwraps every argument and result -- 6-10 us of
    host time per nested op, ~400 of them per training iteration on the autograd thread (tools/host_profile.py, round 5)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoCtx(), *args)
