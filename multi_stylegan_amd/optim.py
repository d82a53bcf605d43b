"""The optimiser steps of the adversarial iteration on flat stores (reference: ``torch.optim.Adam.step()`` at
multi_stylegan/model_wrapper.py:296-300 and :410-414, ``misc.exponential_moving_average`` at :446).

The reference (and round 1 of this package) steps ~400 small parameter tensors per network through torch's Adam.  Its
fused multi-tensor implementation is cheap on the device but costs ~8 ms of HOST time per step -- per-parameter step
counters, pointer tables, one call per parameter group -- and another ~2 ms for the EMA's two ``_foreach`` calls: more
than the device work they enqueue, so the GPU idles behind them (tools/gpu_idle.py, tools/host_profile.py).

``FlatAdam`` keeps the user's ``torch.optim.Adam`` object as the owner of hyper-parameters and state -- ``param_groups``
(learning rates can be changed as usual) and ``state_dict()`` / ``load_state_dict()`` keep the reference's checkpoint
format -- but stores every parameter of a gradient bucket (multi_stylegan_amd.dist.GradBucketReducer), its two moments
and optionally its EMA copy as VIEWS into flat fp32 buffers, and steps a bucket with ONE launch of ``msg_flat_adam``
(csrc/optim.hip; the arithmetic of torch's Adam in the same order).  The gradient-clipping factor and the 1 / world size
of the data-parallel mean ride along as a device scalar, as they did through fused Adam's ``grad_scale``.
"""
from typing import Dict, List, Optional

import torch

from . import _lib


def _hyper(group) -> tuple:
    b1, b2 = group["betas"]
    return (float(group["lr"]), float(b1), float(b2), float(group["eps"]))


def _unshare_steps(_optimizer, state_dict):
    """state_dict post hook: every parameter gets its own copy of its step counter."""
    for entry in state_dict.get("state", {}).values():
        step = entry.get("step")
        if isinstance(step, torch.Tensor):
            entry["step"] = step.clone()
    return state_dict


class FlatAdam:
    """Steps the parameters of ``reducer``'s buckets for ``optimizer`` (a ``torch.optim.Adam``)."""

    @staticmethod
    def supported(optimizer, reducer) -> bool:
        if not isinstance(optimizer, torch.optim.Adam) or type(optimizer) is not torch.optim.Adam:
            return False
        for g in optimizer.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("weight_decay", 0) != 0 or g.get("differentiable") or \
                    isinstance(g["lr"], torch.Tensor):
                return False
        owned = {id(p) for g in optimizer.param_groups for p in g["params"]}
        for b in reducer.buckets:
            if not b.flat.is_cuda or any(id(p) not in owned or p.dtype != torch.float32 for p in b.params):
                return False
        return bool(reducer.buckets)

    def __init__(self, optimizer: torch.optim.Adam, reducer):
        self.optimizer, self.reducer = optimizer, reducer
        self._group_of: Dict[int, dict] = {}
        self._groups_seen: tuple = ()
        self._adopted = False                                             # (the flat stores are still all zeros)
        self._refresh_groups()
        self.step_count = 0
        self._step_tensor = torch.zeros((), dtype=torch.float32)          # what state_dict() shows as every 'step'
        self.param_flat: List[torch.Tensor] = []
        self.exp_avg: List[torch.Tensor] = []
        self.exp_avg_sq: List[torch.Tensor] = []
        self.ema_flat: List[Optional[torch.Tensor]] = []
        self._ema_rest: List[tuple] = []                                  # (ema parameter, parameter) pairs outside the buckets
        self._ema_model = self._ema_train = None
        self._ema_params: List[Optional[List[torch.nn.Parameter]]] = []
        for b in reducer.buckets:
            self.param_flat.append(torch.zeros_like(b.flat))            # (zeros in the alignment padding between parameters)
            self.exp_avg.append(torch.zeros_like(b.flat))
            self.exp_avg_sq.append(torch.zeros_like(b.flat))
            self.ema_flat.append(None)
        self.adopt()
        # state_dict() must not show ONE step tensor shared by every parameter: torch.save keeps the aliasing, and an Adam
        # that advances `step` in place per parameter (CPU parameters, the torch 1.8 the reference pins) would then count
        # ~400 steps per step after loading such a checkpoint
        optimizer.register_state_dict_post_hook(_unshare_steps)
        # someone stepping the torch optimizer directly (not through ModelWrapper._step) gets the state in the form torch's
        # own Adam expects; the next flat step adopts what it did
        optimizer.register_step_pre_hook(lambda _opt, _args, _kwargs: self.release())

    def _refresh_groups(self) -> None:
        """``optimizer.param_groups`` is the owner of the hyper-parameters; ``load_state_dict`` REPLACES the group dicts (and a
        scheduler or the user may edit them), so the parameter -> group table is rebuilt whenever the list's dicts change."""
        seen = tuple(id(g) for g in self.optimizer.param_groups)
        if seen != self._groups_seen:
            self._group_of = {id(p): g for g in self.optimizer.param_groups for p in g["params"]}
            self._groups_seen = seen

    # ---------------------------------------------------------------------------------------------------------
    def adopt(self) -> None:
        """(Re-)establish the flat layout from whatever the parameters and the optimizer's state hold now: after
        construction, after ``load_state_dict`` of the optimizer (which replaces the state tensors) or after anything
        that re-allocated a parameter's storage."""
        self._refresh_groups()
        state = self.optimizer.state
        steps = []
        with torch.no_grad():
            for k, b in enumerate(self.reducer.buckets):
                for p, off in zip(b.params, b.offsets):
                    n = p.numel()
                    view = self.param_flat[k][off:off + n].view_as(p)
                    if p.data.data_ptr() != view.data_ptr():
                        view.copy_(p.data)
                        p.data = view
                    st = state.get(p, None)
                    m_view = self.exp_avg[k][off:off + n].view_as(p)
                    v_view = self.exp_avg_sq[k][off:off + n].view_as(p)
                    if st:
                        if st["exp_avg"].data_ptr() != m_view.data_ptr():
                            m_view.copy_(st["exp_avg"])
                            v_view.copy_(st["exp_avg_sq"])
                        steps.append(int(float(st["step"])))
                    elif self._adopted:
                        # no moments came with this parameter (a checkpoint that lacks it): its slices of the flat stores
                        # still hold the previous run's moments
                        m_view.zero_()
                        v_view.zero_()
                    state[p] = {"step": self._step_tensor, "exp_avg": m_view, "exp_avg_sq": v_view}
        if steps:
            # torch counts per parameter; every parameter of the buckets is stepped together here.  (A checkpoint
            # whose parameters disagree -- some never received a gradient -- resumes at the largest count.)
            self.step_count = max(steps)
        self._step_tensor.fill_(float(self.step_count))
        self._adopted = True

    def release(self) -> None:
        """Hand the state back in the form torch's own (fused) Adam expects -- a step counter per parameter on its
        device -- before a step that goes through ``optimizer.step()``.  The moments stay views of the flat stores; the
        next flat step adopts whatever torch did."""
        for b in self.reducer.buckets:
            for p in b.params:
                st = self.optimizer.state.get(p)
                if st and st["step"] is self._step_tensor:
                    st["step"] = torch.tensor(float(self.step_count), dtype=torch.float32, device=p.device)

    def _layout_intact(self) -> bool:
        """Every parameter still IS its slice of the flat store (a re-pointed `.data` would keep computing with detached
        storage while msg_flat_adam updates the store), and the first parameter's state is the flat state (load_state_dict
        replaces all state entries together).  ~400 integer compares per step."""
        state = self.optimizer.state
        for k, b in enumerate(self.reducer.buckets):
            base = self.param_flat[k].data_ptr()
            for p, off in zip(b.params, b.offsets):
                if p.data.data_ptr() != base + 4 * off:
                    return False
            st = state.get(b.params[0])
            if not st or st["step"] is not self._step_tensor or st["exp_avg"].data_ptr() != self.exp_avg[k].data_ptr():
                return False
        return True

    def _bucket_hyper(self, b) -> Optional[tuple]:
        hyper = _hyper(self._group_of[id(b.params[0])])
        seen = {id(self._group_of[id(b.params[0])])}
        for p in b.params:
            g = self._group_of[id(p)]
            if id(g) not in seen:
                seen.add(id(g))
                if _hyper(g) != hyper:
                    return None
        return hyper

    # ---------------------------------------------------------------------------------------------------------
    def step(self, coef: Optional[torch.Tensor] = None) -> bool:
        """One Adam update of every bucket; ``coef`` (device scalar, fp32) multiplies the gradients first.  Returns False
        -- nothing done -- if a bucket mixes parameter groups whose hyper-parameters differ (the caller then steps
        through torch)."""
        from . import conv_ops
        self._refresh_groups()
        hypers = [self._bucket_hyper(b) for b in self.reducer.buckets]
        if any(h is None for h in hypers):
            return False
        if not self._layout_intact():
            self.adopt()
        self.step_count += 1
        self._step_tensor.fill_(float(self.step_count))
        dev = self.reducer.buckets[0].flat.device
        if coef is not None:
            coef = coef.detach().to(dev, torch.float32).reshape(1)
        with _lib.on_device(dev):
            stream = _lib.stream_of(dev)
            for k, b in enumerate(self.reducer.buckets):
                lr, b1, b2, eps = hypers[k]
                n = b.flat.numel()
                with _lib.kernel_clock.span("flat_adam/f32", 28 * n):
                    code = _lib.lib().msg_flat_adam(self.param_flat[k].data_ptr(), b.flat.data_ptr(),
                                                    self.exp_avg[k].data_ptr(), self.exp_avg_sq[k].data_ptr(), None, n,
                                                    _lib.ptr(coef), lr, b1, b2, eps, self.step_count, 0.0, stream)
                _lib.check(code, "msg_flat_adam")
        # the parameters changed behind autograd's back: cached kernel-side weight images are stale
        conv_ops.invalidate_weight_cache([p for b in self.reducer.buckets for p in b.params])
        return True

    # ---------------------------------------------------------------------------------------------------------
    def attach_ema(self, model_ema: torch.nn.Module, model_train: torch.nn.Module) -> None:
        """Give the EMA copy's parameters the same flat layout, so that its update is one launch per bucket."""
        names = {id(p): n for n, p in model_train.named_parameters()}
        ema_named: Dict[str, torch.nn.Parameter] = dict(model_ema.named_parameters())
        covered = set()
        with torch.no_grad():
            for k, b in enumerate(self.reducer.buckets):
                if any(id(p) not in names or names[id(p)] not in ema_named or
                       ema_named[names[id(p)]].shape != p.shape or ema_named[names[id(p)]].dtype != torch.float32 or
                       ema_named[names[id(p)]].device != p.device for p in b.params):
                    continue
                flat = torch.zeros_like(b.flat)
                for p, off in zip(b.params, b.offsets):
                    e = ema_named[names[id(p)]]
                    view = flat[off:off + p.numel()].view_as(e)
                    view.copy_(e.data)
                    e.data = view
                    covered.add(names[id(p)])
                self.ema_flat[k] = flat
        train_named = dict(model_train.named_parameters())
        self._ema_rest = [(e, train_named[n]) for n, e in ema_named.items() if n not in covered]
        self._ema_model, self._ema_train = model_ema, model_train
        self._ema_params = [None if self.ema_flat[k] is None else [ema_named[names[id(p)]] for p in b.params]
                            for k, b in enumerate(self.reducer.buckets)]

    def _ema_intact(self) -> bool:
        for k, flat in enumerate(self.ema_flat):
            if flat is None:
                continue
            base = flat.data_ptr()
            for e, off in zip(self._ema_params[k], self.reducer.buckets[k].offsets):
                if e.data.data_ptr() != base + 4 * off:
                    return False
        return True

    def ema_update(self, decay: float = 0.999) -> None:
        """ema <- decay * ema + (1 - decay) * parameter (misc.exponential_moving_average) on the flat stores."""
        from . import conv_ops
        if self._ema_model is None:
            raise RuntimeError("FlatAdam.ema_update: attach_ema first")
        if not self._layout_intact() or not self._ema_intact():
            self.adopt()
            self.attach_ema(self._ema_model, self._ema_train)
        dev = self.reducer.buckets[0].flat.device
        with torch.no_grad(), _lib.on_device(dev):
            stream = _lib.stream_of(dev)
            for k, b in enumerate(self.reducer.buckets):
                if self.ema_flat[k] is None:
                    continue
                n = b.flat.numel()
                with _lib.kernel_clock.span("flat_ema/f32", 12 * n):
                    code = _lib.lib().msg_flat_ema(self.ema_flat[k].data_ptr(), self.param_flat[k].data_ptr(), n,
                                                   float(decay), stream)
                _lib.check(code, "msg_flat_ema")
            if self._ema_rest:
                ema = [e.data for e, _ in self._ema_rest]
                torch._foreach_mul_(ema, decay)
                torch._foreach_add_(ema, [p.data for _, p in self._ema_rest], alpha=1 - decay)
        conv_ops.invalidate_weight_cache(list(self._ema_model.parameters()))
