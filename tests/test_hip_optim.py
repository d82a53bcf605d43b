"""msg_flat_adam / msg_flat_ema through multi_stylegan_amd.optim.FlatAdam against torch.optim.Adam (the optimiser the
reference steps with, model_wrapper.py:296-300 / :410-414) and against misc.exponential_moving_average."""
import copy

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(3, 5), (7,), (16, 16, 3, 3), (1,), (33,), (2, 129)]          # 2 641 elements: not a multiple of four


def _setup(split=True, seed=0):
    from multi_stylegan_amd import dist as msg_dist, optim as msg_optim
    torch.manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in SHAPES]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    groups = lambda ps: [{"params": ps[:4], "lr": 2e-3}, {"params": ps[4:], "lr": 5e-5}]
    opt = torch.optim.Adam(groups(params), betas=(0.0, 0.99), fused=True)
    opt_ref = torch.optim.Adam(groups(ref), betas=(0.0, 0.99))
    key = {id(p): g["lr"] for g in opt.param_groups for p in g["params"]}
    red = msg_dist.GradBucketReducer(params, split_key=(lambda p: key[id(p)]) if split else None)
    return params, ref, opt, opt_ref, red, msg_optim


def test_flat_adam_matches_torch_adam():
    params, ref, opt, opt_ref, red, msg_optim = _setup()
    assert len(red.buckets) == 2 and msg_optim.FlatAdam.supported(opt, red)
    flat = msg_optim.FlatAdam(opt, red)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for step in range(1, 7):
        coef = torch.rand((), device=DEV, generator=gen) + 0.25
        for p, q in zip(params, ref):
            g = torch.randn(p.shape, device=DEV, generator=gen) * (10.0 ** (step % 3 - 1))
            p.grad.copy_(g)                                   # (views into the reducer's flat buckets)
            q.grad = g * coef
        if step == 4:                                         # learning rates are read from param_groups at every step
            for o in (opt, opt_ref):
                o.param_groups[0]["lr"] = 7e-4
        assert flat.step(coef)
        opt_ref.step()
        for p, q in zip(params, ref):
            assert rel_err(p.detach(), q.detach()) < 4e-6, step
            assert rel_err(opt.state[p]["exp_avg_sq"], opt_ref.state[q]["exp_avg_sq"]) < 4e-6
            assert rel_err(opt.state[p]["exp_avg"], opt_ref.state[q]["exp_avg"]) < 4e-6
            assert float(opt.state[p]["step"]) == float(opt_ref.state[q]["step"]) == step
    # the state dict is torch.optim.Adam's: it loads into a plain Adam, which then continues identically
    fresh = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt2 = torch.optim.Adam([{"params": fresh[:4], "lr": 1.0}, {"params": fresh[4:], "lr": 1.0}], betas=(0.5, 0.5))
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    for p, q, f in zip(params, ref, fresh):
        g = torch.randn(p.shape, device=DEV, generator=gen)
        p.grad.copy_(g)
        q.grad = g.clone()
        f.grad = g.clone()
    assert flat.step(None)
    opt_ref.step()
    opt2.step()
    for p, q, f in zip(params, ref, fresh):
        assert rel_err(p.detach(), q.detach()) < 4e-6 and rel_err(f.detach(), q.detach()) < 4e-6


def test_flat_adam_adopts_loaded_state_and_reallocated_parameters():
    params, ref, opt, opt_ref, red, msg_optim = _setup(seed=3)
    flat = msg_optim.FlatAdam(opt, red)
    gen = torch.Generator(device=DEV).manual_seed(2)

    def both_step():
        for p, q in zip(params, ref):
            g = torch.randn(p.shape, device=DEV, generator=gen)
            p.grad.copy_(g)
            q.grad = g.clone()
        assert flat.step(None)
        opt_ref.step()
    both_step()
    both_step()
    # a state dict written by torch's own Adam (the reference's checkpoints) replaces the state tensors ...
    opt.load_state_dict(copy.deepcopy(opt_ref.state_dict()))
    # ... and something re-allocates a parameter behind the flat store's back
    with torch.no_grad():
        params[2].data = params[2].data.clone()
    both_step()
    assert flat.step_count == 3
    for p, q in zip(params, ref):
        assert rel_err(p.detach(), q.detach()) < 4e-6
        assert p.data.data_ptr() >= flat.param_flat[0].data_ptr() or p.data.data_ptr() >= flat.param_flat[1].data_ptr()


def test_flat_adam_declines_mixed_hyperparameters_and_torch_takes_over():
    params, ref, opt, opt_ref, red, msg_optim = _setup(split=False)
    assert len(red.buckets) == 1
    flat = msg_optim.FlatAdam(opt, red)
    for p, q in zip(params, ref):
        p.grad.copy_(torch.ones_like(p))
        q.grad = torch.ones_like(q)
    before = [p.detach().clone() for p in params]
    assert flat.step(None) is False                          # two learning rates in one bucket
    assert all(torch.equal(p.detach(), b) for p, b in zip(params, before))
    opt.step()               # torch's fused Adam on the same (flat-view) state; FlatAdam's step pre-hook hands it over
    opt_ref.step()
    for p, q in zip(params, ref):
        assert rel_err(p.detach(), q.detach()) < 4e-6


def test_flat_ema_matches_reference_formula():
    from multi_stylegan_amd import dist as msg_dist, misc, optim as msg_optim
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(9, 17), torch.nn.Linear(17, 3)).to(DEV)
    ema = copy.deepcopy(net)
    with torch.no_grad():
        for p in ema.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    ema_ref, net_ref = copy.deepcopy(ema), copy.deepcopy(net)
    live = list(net.parameters())[:3]                         # the last parameter is not in any bucket: foreach remainder
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    red = msg_dist.GradBucketReducer(live)
    flat = msg_optim.FlatAdam(opt, red)
    flat.attach_ema(ema, net)
    for _ in range(3):
        with torch.no_grad():
            for p, q in zip(net.parameters(), net_ref.parameters()):
                step = torch.randn_like(p) * 0.05
                p.add_(step)
                q.add_(step)
        flat.ema_update(0.9)
        misc.exponential_moving_average(ema_ref, net_ref, 0.9)
    for (n, p), q in zip(ema.named_parameters(), ema_ref.parameters()):
        assert rel_err(p.detach(), q.detach()) < 1e-6, n


def test_flat_adam_reads_hyperparameters_of_a_loaded_state_dict():
    """``Optimizer.load_state_dict`` replaces the ``param_groups`` dicts: the checkpoint's learning rates -- and any later
    ``param_groups[i]['lr'] = ...`` or scheduler step, which edit the NEW dicts -- must be what the flat step uses (round-2
    advice: the parameter -> group table was captured once and went stale)."""
    params, ref, opt, opt_ref, red, msg_optim = _setup(seed=7)
    flat = msg_optim.FlatAdam(opt, red)
    gen = torch.Generator(device=DEV).manual_seed(3)

    def both_step():
        for p, q in zip(params, ref):
            g = torch.randn(p.shape, device=DEV, generator=gen)
            p.grad.copy_(g)
            q.grad = g.clone()
        assert flat.step(None)
        opt_ref.step()
    both_step()
    old_groups = [id(g) for g in opt.param_groups]
    sd = copy.deepcopy(opt_ref.state_dict())
    sd["param_groups"][0]["lr"], sd["param_groups"][1]["lr"] = 9e-3, 4e-4          # the checkpoint's own learning rates
    opt.load_state_dict(copy.deepcopy(sd))
    opt_ref.load_state_dict(copy.deepcopy(sd))
    assert [id(g) for g in opt.param_groups] != old_groups                          # torch did replace the dicts
    both_step()
    for p, q in zip(params, ref):
        assert rel_err(p.detach(), q.detach()) < 4e-6
    for o in (opt, opt_ref):                                                         # a scheduler-style edit afterwards
        o.param_groups[0]["lr"] = 1e-4
    both_step()
    for p, q in zip(params, ref):
        assert rel_err(p.detach(), q.detach()) < 4e-6


def test_flat_adam_state_dict_has_one_step_tensor_per_parameter_and_notices_any_repointed_parameter():
    params, ref, opt, opt_ref, red, msg_optim = _setup(seed=8)
    flat = msg_optim.FlatAdam(opt, red)
    gen = torch.Generator(device=DEV).manual_seed(4)

    def both_step():
        for p, q in zip(params, ref):
            g = torch.randn(p.shape, device=DEV, generator=gen)
            p.grad.copy_(g)
            q.grad = g.clone()
        assert flat.step(None)
        opt_ref.step()
    both_step()
    steps = [st["step"] for st in opt.state_dict()["state"].values()]
    assert len({s.data_ptr() for s in steps}) == len(steps) and all(float(s) == 1.0 for s in steps)
    # a plain (foreach / single-tensor) Adam that advances `step` in place per parameter continues at 2, not at 1 + #params
    fresh = [torch.nn.Parameter(p.detach().clone().cpu()) for p in params]
    opt_cpu = torch.optim.Adam([{"params": fresh[:4], "lr": 2e-3}, {"params": fresh[4:], "lr": 5e-5}], betas=(0.0, 0.99))
    opt_cpu.load_state_dict(copy.deepcopy(opt.state_dict()))
    for f in fresh:
        f.grad = torch.zeros_like(f)
    opt_cpu.step()
    assert all(float(st["step"]) == 2.0 for st in opt_cpu.state.values())
    # re-pointing a parameter that is NOT the first of its bucket, with no load_state_dict around it
    with torch.no_grad():
        params[1].data = params[1].data.clone()
    assert not flat._layout_intact()
    both_step()
    assert flat._layout_intact()
    for p, q in zip(params, ref):
        assert rel_err(p.detach(), q.detach()) < 4e-6
