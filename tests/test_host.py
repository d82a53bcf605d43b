"""CPU-side checks of the product's host logic: C-ABI library exports, module/state_dict surface, losses, the
gradient-bucket reducer and the trainer's data-parallel path under gloo (world size 2)."""
import ctypes
import json
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, rel_err


def test_library_exports_every_declared_symbol():
    from multi_stylegan_amd import _lib
    from multi_stylegan_amd.build import build
    build(verbose=False)
    names = _lib.declared_symbols()
    assert {"msg_upfirdn2d", "msg_fused_bias_act", "msg_bias_act_backward", "msg_abi_version"} <= set(names)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/msg_hip.h but not exported"
    assert _lib.lib().msg_build_arch() == b"gfx950" and _lib.lib().msg_abi_version() >= 1
    assert set(_lib._SIGNATURES) == set(names)            # the ctypes table binds all of them, nothing else


def test_product_modules_keep_reference_surface():
    import multi_stylegan_amd as m
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    g = m.MultiStyleGANGenerator(m.multi_style_gan_generator_config)
    d = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    assert {k: list(v.shape) for k, v in g.state_dict().items()} == man["generator"]
    assert {k: list(v.shape) for k, v in d.state_dict().items()} == man["discriminator"]
    assert [len(list(grp["params"])) for grp in g.get_parameters()] == man["generator_param_groups"]
    assert sum(p.numel() for p in g.live_parameters()) == 32565276
    with pytest.raises(Exception, match="no CPU fallback"):
        g(torch.randn(2, 512))                                           # product path refuses CPU tensors


def test_losses_match_oracle(golden):
    from multi_stylegan_amd import loss
    from oracle import train as ot
    torch.manual_seed(0)
    pr, pf = torch.randn(4, 1), torch.randn(4, 1, 1, 8, 8)
    assert rel_err(loss.NonSaturatingLogisticGeneratorLoss()(pf), ot.g_logistic_loss(pf)) < 1e-6
    a, b = loss.NonSaturatingLogisticDiscriminatorLoss()(pr, pr * 2)
    c, d_ = ot.d_logistic_loss(pr, pr * 2)
    assert rel_err(a, c) < 1e-6 and rel_err(b, d_) < 1e-6
    # path length: value, running mean and -- the subtle part -- the gradient through the running mean
    pl_a, pl_b = loss.PathLengthRegularization(), ot.PathLength()
    for _ in range(3):
        g1 = torch.randn(2, 5, 16, requires_grad=True)
        g2 = g1.detach().clone().requires_grad_(True)
        la, _ = pl_a(g1)
        lb, _ = pl_b(g2)
        la.backward(); lb.backward()
        assert rel_err(la, lb) < 1e-6 and rel_err(g1.grad, g2.grad) < 1e-6
    assert rel_err(pl_a.mean_path_length, pl_b.mean_path_length) < 1e-6


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 3)
        self.unused = torch.nn.Parameter(torch.ones(4))

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _reducer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_stylegan_amd.dist import GradBucketReducer
    model = _Toy()
    params = [p for n, p in model.named_parameters() if n != "unused"]
    red = GradBucketReducer(params, bucket_bytes=64, overlap=True)       # tiny buckets -> several of them
    assert len(red.buckets) > 1
    torch.manual_seed(100 + rank)
    x = torch.randn(7, 6)
    for attempt in range(2):                                             # second pass exercises zero_grad reuse
        local = [t.clone() for t in torch.autograd.grad(model(x).square().sum(), params)]   # before any collective
        red.zero_grad(); red.arm()
        model(x).square().sum().backward()                 # buckets are all-reduced from the hooks, overlapped
        red.finish()
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        for i, p in enumerate(params):
            want = sum(g[i] for g in gathered) / world
            assert torch.allclose(p.grad, want, atol=1e-6), (rank, attempt, i)
        total = red.clip_(0.5)
        norm_after = torch.sqrt(sum(p.grad.square().sum() for p in params))
        assert norm_after <= 0.5 + 1e-4 and total > 0
    # gradients arriving in a different order on each rank must not reorder the collectives
    local = [t.clone() for t in torch.autograd.grad(model(x).square().sum(), params)]
    red.zero_grad(); red.arm()
    # (rank r starts its arrival order at a different parameter and odd ranks walk it backwards: four distinct orders at world 4)
    order = [(i + rank) % len(params) for i in range(len(params))]
    if rank % 2:
        order.reverse()
    for i in order:
        params[i].grad.copy_(local[i])
        red._on_grad(params[i])
    red.finish()
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    for i, p in enumerate(params):
        assert torch.allclose(p.grad, sum(g[i] for g in gathered) / world, atol=1e-6), (rank, "order", i)
    assert model.unused.grad is None
    # ---- labelled backwards: a bucket that a kind of backward never fills must not stall the buckets behind it
    red2 = GradBucketReducer(params + [model.unused], bucket_bytes=16, overlap=True)      # one parameter per bucket,
    cold = next(k for k, b in enumerate(red2.buckets) if b.params[0] is model.unused)     # the never-filled one FIRST
    assert cold == 0 and all(len(b.params) == 1 for b in red2.buckets) and len(red2.buckets) == len(params) + 1
    for attempt in range(3):
        local = [t.clone() for t in torch.autograd.grad(model(x).square().sum(), params)]
        red2.zero_grad(); red2.arm("fwd")
        model(x).square().sum().backward()
        launched = [b.work is not None for b in red2.buckets]          # before finish(): what the hooks managed to send
        if attempt == 0:
            # first time nothing is known: launches are strictly in order and every bucket waits behind the cold one
            assert not any(launched), launched
        else:
            assert not launched[cold] and all(launched[1:]), (attempt, launched)
        red2.finish()
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        for i, p in enumerate(params):
            assert torch.allclose(p.grad, sum(g[i] for g in gathered) / world, atol=1e-6), (rank, "labelled", attempt, i)
        assert float(model.unused.grad.abs().max()) == 0.0             # exchanged (last), still zero
    # a cold bucket that does receive a gradient after all is simply exchanged last, with that gradient
    red2.zero_grad(); red2.arm("fwd")
    (model(x).square().sum() + (1.0 + rank) * model.unused.sum()).backward()
    red2.finish()
    assert torch.allclose(model.unused.grad, torch.full((4,), 1.0 + (world - 1) / 2.0))       # mean of 1 + rank
    # ... but a gradient that arrives for a bucket ALREADY on the wire is refused loudly, never silently dropped
    red4 = GradBucketReducer(params + [model.unused], bucket_bytes=64, overlap=True)
    mixed = next(b for b in red4.buckets if any(q is model.unused for q in b.params))
    assert len(mixed.params) > 1
    for _ in range(2):
        red4.zero_grad(); red4.arm("fwd")
        model(x).square().sum().backward()
        red4.finish()
    red4.zero_grad(); red4.arm("fwd")
    try:
        (model(x).square().sum() + model.unused.sum()).backward()
        raised = False
    except RuntimeError as exc:
        raised = "use distinct labels" in str(exc)
    red4.finish()
    assert raised
    for p in params:                                                    # hand the gradients back to the first reducer
        p.grad = None
    model.unused.grad = None
    # ---- reduce_scatter exchange: same means, same clip norm as the all-reduce exchange
    red3 = GradBucketReducer(params, bucket_bytes=64, overlap=True, exchange="reduce_scatter")
    local = [t.clone() for t in torch.autograd.grad(model(x).square().sum(), params)]
    red3.zero_grad(); red3.arm("fwd")
    model(x).square().sum().backward()
    red3.finish()
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    means = [sum(g[i] for g in gathered) / world for i in range(len(params))]
    for i, p in enumerate(params):
        assert torch.allclose(p.grad, means[i], atol=1e-6), (rank, "reduce_scatter", i)
    want_norm = torch.sqrt(sum(m.square().sum() for m in means))
    assert torch.allclose(red3.grad_norm(), want_norm, rtol=1e-5), (red3.grad_norm(), want_norm)
    red3.clip_(0.5)
    assert torch.sqrt(sum(p.grad.square().sum() for p in params)) <= 0.5 + 1e-4
    # ... and the SUM form the fused optimiser path uses: finish(average=False) leaves sums, the norm is the sum's
    red3.zero_grad(); red3.arm("fwd")
    model(x).square().sum().backward()
    inv = red3.finish(average=False)
    assert inv == 1.0 / world and torch.allclose(red3.grad_norm() * inv, want_norm, rtol=1e-5)
    for i, p in enumerate(params):
        assert torch.allclose(p.grad * inv, means[i], atol=1e-6)
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def _run_ranks(worker, world, port, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(timeout) for p in procs]
    hung = [p for p in procs if p.exitcode is None]
    [p.kill() for p in hung]
    assert not hung, "ranks deadlocked"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


@pytest.mark.parametrize("world", [2, 4])
def test_bucket_reducer_gloo(world):
    """world 4: four distinct gradient-arrival orders (collectives must still be issued in ONE order), means over four
    shards, both exchanges."""
    _run_ranks(_reducer_worker, world, 29000 + os.getpid() % 400 + 17 * world, 180)


def _trainer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from multi_stylegan_amd.model_wrapper import ModelWrapper
    from oracle import models as om                      # CPU stand-ins for G/D: the trainer logic is device-agnostic
    from tools.gen_golden import TINY_D, TINY_G
    torch.set_num_threads(max(1, 8 // world))
    torch.manual_seed(10 + rank)                         # different init per rank: broadcast must fix it
    g, d = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    g.live_parameters = lambda: [p for n, p in g.named_parameters() if not n.startswith("main_convolutions_2.")]
    orig_forward = g.forward
    g.forward = lambda *a, path_length_noise=None, **k: orig_forward(*a, **k)
    # plain SGD: the parameter movement is proportional to the exchanged gradient, so a wrong reduction shows
    tr = ModelWrapper(g, d, device="cpu", bucket_bytes=1 << 16,
                      generator_optimizer=torch.optim.SGD(g.parameters(), lr=1e-3),
                      discriminator_optimizer=torch.optim.SGD(d.parameters(), lr=1e-3))
    from ddp_probe import StepProbe
    probe = StepProbe(tr)
    tr.iteration = 15                                    # next iteration is 16: R1 and path length fire
    torch.manual_seed(1000 + rank)                       # different data per rank
    # rank-distinct style-mixing graphs (what a real job has: every rank draws its own mixing coin and crossover layer):
    # different crossover layers per rank, and the last rank of a world > 2 does not mix at all -- its generator graph has
    # ONE mapping-network pass where the others have two, so gradients become ready in different orders on different ranks
    import random
    import numpy as np
    random.seed(4321 + rank)
    np.random.seed(99 + 7 * rank)
    if world > 2 and rank == world - 1:
        tr.hyperparameters = dict(tr.hyperparameters, p_mixed_noise=0.0)
    tr.train_iteration(torch.rand(2, 2, 3, 32, 32))
    logs = tr.pop_logs()
    # every step == the single-process step with the mean of the two shards' gradients
    assert probe.check(world, lr=1e-3) == ["d", "g", "pl", "r1"]
    assert {"loss_discriminator_regularization", "path_length", "loss_generator"} <= set(logs)
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], other) for other in both[1:]), "replicas diverged"
    mpl = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(mpl, tr.path_length_regularization.mean_path_length)
    assert all(torch.equal(mpl[0], other) for other in mpl[1:])
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_trainer_data_parallel_gloo(world):
    """One regularised iteration (D, R1, G, path-length steps) on `world` gloo ranks with rank-distinct data, latents and
    style-mixing graphs: every optimiser step is the step of the MEAN of the shards' gradients (tests/ddp_probe.py), the
    replicas and the path-length mean stay bit-identical."""
    _run_ranks(_trainer_worker, world, 29500 + os.getpid() % 300 + 23 * world, 420)


def _cut_mix_gate_worker(rank, world, port, out):
    """Late-training iterations with the RANDOM CutMix gate (no Draws.cut_mix) and rank-distinct Python RNGs, as bench.py
    and the per-rank style-mixing draws leave them: every rank must take the same branch in every iteration."""
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from multi_stylegan_amd.model_wrapper import ModelWrapper
    from oracle import models as om
    from tools.gen_golden import TINY_D, TINY_G
    torch.manual_seed(10)
    g, d = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    g.live_parameters = lambda: [p for n, p in g.named_parameters() if not n.startswith("main_convolutions_2.")]
    orig_forward = g.forward
    g.forward = lambda *a, path_length_noise=None, **k: orig_forward(*a, **k)
    random.seed(1234 + rank)
    tr = ModelWrapper(g, d, device="cpu", bucket_bytes=1 << 16,
                      generator_optimizer=torch.optim.SGD(g.parameters(), lr=1e-3),
                      discriminator_optimizer=torch.optim.SGD(d.parameters(), lr=1e-3))
    assert tr._control_rng is not None
    tr.epoch, tr.epochs = 5, 10                          # gate probability 0.25, plus the resume_training coin
    torch.manual_seed(1000 + rank)
    gates = []
    for _ in range(4):
        tr.train_iteration(torch.rand(2, 2, 3, 32, 32), resume_training=True)
        gates.append("loss_cut_mix_augmentation" in tr.pop_logs())
    everyone = [None] * world
    dist.all_gather_object(everyone, gates)
    assert everyone[0] == everyone[1], everyone
    assert any(gates), gates                             # the CutMix branch (two extra discriminator exchanges) was taken
    mine = [None] * world                                # ... while the ranks' own Python RNGs really are distinct
    dist.all_gather_object(mine, random.random())
    assert mine[0] != mine[1]
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    # a checkpoint carries the gate's generator state: a resumed job draws the same continuation on every rank
    state = tr.checkpoint_dict()["multi_stylegan_amd"]["control_rng"]
    nxt = tr._control_random()
    tr._control_rng.setstate(state)
    assert tr._control_random() == nxt
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_cut_mix_gate_is_rank_consistent_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28200 + os.getpid() % 400
    procs = [ctx.Process(target=_cut_mix_gate_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    hung = [p for p in procs if p.exitcode is None]
    [p.kill() for p in hung]
    assert not hung, "ranks disagreed on the CutMix gate and deadlocked"
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"


def test_kernel_plans_respect_the_31_bit_offset_limits():
    """Host-side dispatch only (no launch): the kernels that address their operands through buffer descriptors -- 31-bit
    offsets -- must not be planned for tensors of 2 GiB or more (the launch then belongs to the 64-bit-pointer instantiation),
    and the weight-gradient planner must accept the maps one column short of a power of two that it now runs on padded rows.
    The GPU side of the same limits: tests/test_hip_conv.py::test_conv_above_two_gib_of_activations."""
    from multi_stylegan_amd import _lib
    from multi_stylegan_amd.build import build
    build(verbose=False)
    lib = _lib.lib()
    bf16 = _lib.MSG_BF16
    plan = lambda b, c, n, hw, k, ws: lib.msg_conv2d_fprop_plan(bf16, b, hw, hw, c, c, hw, hw, n, k, k, ws)
    # 3x3 512 -> 512 @256^2: per-sample weights (one sample per descriptor) and a shared-weight batch of 0.5 GiB: the 256 x 256 tile
    assert plan(16, 512, 512, 256, 3, 512 * 512 * 9) == 3
    assert plan(8, 512, 512, 256, 3, 0) == 3
    # the same layer over 33 samples with SHARED weights: 2.2 GiB behind one descriptor -> register-staged 128 x 128 kernel
    assert plan(33, 512, 512, 256, 3, 0) == 0
    # 1x1 512 -> 256 (ping-pong kernel's territory: plan 2) likewise
    assert plan(16, 512, 256, 256, 1, 0) == 2
    assert plan(33, 512, 256, 256, 1, 0) == 0
    # weight-gradient workspace queries (plan only) on the discriminator's stride-2 outputs: 127, 63, 31, 15 wide
    for c, ihw, ohw in ((128, 256, 127), (256, 128, 63), (384, 64, 31), (768, 32, 15)):
        need = lib.msg_conv2d_wgrad_workspace(bf16, 32, ihw, ihw, c, c, ohw, ohw, c, c, c, 3, 3, 2, 0, 0, 0, 1)
        assert need > 0 and need % (c * 9 * c) == 0, (c, ohw, need)          # whole slabs of O x taps x ldgw floats


def test_non_square_conv_geometry_is_refused():
    """EqualizedConv2d keeps the reference's (h, w) tuple arguments; a non-square stride / padding must raise instead
    of being computed with the first entry (round-1 advice)."""
    from multi_stylegan_amd import _lib, conv_ops
    assert conv_ops._square((2, 2), "stride") == 2 and conv_ops._square(1, "padding") == 1
    for bad in ((1, 2), (2, 1), (1, 1, 1)):
        with pytest.raises(_lib.MsgHipError, match="square"):
            conv_ops._square(bad, "stride")


# ------------------------------------------------------------------------------ trainer surface, host logic
def _cpu_trainer(golden, **kw):
    """The product's ModelWrapper driving the CPU oracle's modules: the trainer's control flow (step order, zeroing,
    late-training branches, top-k, clipping, EMA, checkpoints) is device-agnostic and is checked here without a GPU;
    the same iterations run on the HIP modules in tests/test_hip_models.py."""
    import copy
    from multi_stylegan_amd.model_wrapper import ModelWrapper
    from oracle import models as om
    from tools.gen_golden import TINY_D, TINY_G
    z = golden("train_step")
    g, d = om.Generator(TINY_G), om.Discriminator(TINY_D, no_rfp=True)
    g.load_state_dict(z.state_dict("train.G0.")); d.load_state_dict(z.state_dict("train.D0."))
    ema = copy.deepcopy(g)
    ema.load_state_dict(z.state_dict("train.Gema0."))
    for mod in (g, ema):
        mod.live_parameters = lambda mod=mod: [p for n, p in mod.named_parameters()
                                               if not n.startswith("main_convolutions_2.")]
    orig_forward = g.forward

    def forward(*a, path_length_noise=None, **k):            # the oracle draws the image noise inside (generator.py:195)
        if path_length_noise is None:
            return orig_forward(*a, **k)
        from oracle.train import _pl_grads
        k.pop("return_path_length_grads")
        return _pl_grads(g, k.pop("input"), k.pop("inject_index"), k.pop("noise"), path_length_noise)
    g.forward = forward
    return z, g, d, ModelWrapper(g, d, generator_ema=ema, device="cpu", **kw)


def test_trainer_three_golden_iterations_on_cpu(golden):
    """ModelWrapper.train_iteration (host logic) against the reference-driven golden run, iteration 32 included: wrongly
    ordered reals, CutMix augmentation + consistency, top-k with v = 0.5 -- every optimiser step whole."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import check_step_trace
    from multi_stylegan_amd import loss, model_wrapper
    from test_oracle_golden import GOLDEN_ITERATIONS, STEP_LABELS, load_train_draws, split_trace, step_traces
    z, g, d, tr = _cpu_trainer(golden)
    top_k = loss.TopK(0, 1)
    names = {"cut_mix_aug": "loss_cut_mix_augmentation", "cut_mix_reg": "loss_cut_mix_regularization",
             "loss_g": "loss_generator", "r1": "loss_discriminator_regularization", "loss_pl": "loss_path_length_regularization"}
    for step, (iteration, late) in enumerate(GOLDEN_ITERATIONS):
        real, draws = load_train_draws(z, step, model_wrapper)
        tr.iteration = iteration - 1
        tr.step_trace = {}
        tr.train_iteration(real, draws, resume_training=late, top_k=top_k if late else None)
        log = tr.pop_logs()
        pre = f"train.it{step}."
        want_steps, want_ema = step_traces(z, pre)
        got_steps, got_ema = split_trace(tr.step_trace)
        assert list(got_steps) == STEP_LABELS[iteration]
        for label, want in want_steps.items():
            st = check_step_trace(got_steps[label], want, tol_grad=5e-4, tol_norm=1e-4, tol_delta=2e-3)
            assert st["compared"] > 0.2 * st["total"], (label, st)
        for n, want in want_ema.items():
            assert rel_err(got_ema[n], want) < 2e-3, n
        for short, long in names.items():
            if z.keys(pre + "log." + short):
                want = float(z[pre + "log." + short])
                assert abs(log[long][0] - want) <= 2e-4 * abs(want), (short, log[long][0], want)


def test_checkpoint_round_trip_and_reference_layout(golden, tmp_path):
    """save_checkpoint writes the reference's six entries (model_wrapper.py:181-192); load_checkpoint restores a run
    exactly, and also accepts the reference's own layouts: DataParallel `module.` prefixes, the ADA wrapper's
    `discriminator.` prefix, an empty path-length entry."""
    from multi_stylegan_amd import model_wrapper
    from test_oracle_golden import load_train_draws
    z, g, d, tr = _cpu_trainer(golden)
    real, draws = load_train_draws(z, 1, model_wrapper)
    tr.iteration = 15
    tr.train_iteration(real, draws)                           # both regularisers fire: optimiser state, PL mean
    path = str(tmp_path / "models" / "checkpoint_5.pt")
    tr.save_checkpoint(path)
    ck = torch.load(path, weights_only=False)
    assert list(ck)[:6] == ["generator_ema", "generator", "generator_optimizer", "discriminator",
                            "discriminator_optimizer", "path_length_regularization"]
    assert float(ck["path_length_regularization"]["mean_path_length"]) == float(tr.path_length_regularization.mean_path_length)
    _, g2, d2, tr2 = _cpu_trainer(golden)
    tr2.load_checkpoint(path)
    assert tr2.iteration == 16
    for a, b in ((g, g2), (d, d2), (tr.generator_ema, tr2.generator_ema)):
        for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(p, q), n
    assert torch.equal(tr.path_length_regularization.mean_path_length, tr2.path_length_regularization.mean_path_length)
    # the resumed run continues identically (optimiser moments restored, gradients still inside the flat buckets)
    real, draws = load_train_draws(z, 0, model_wrapper)
    for t in (tr, tr2):
        t.train_iteration(real, draws)
    for p, q in zip(list(g.parameters()) + list(d.parameters()), list(g2.parameters()) + list(d2.parameters())):
        assert torch.equal(p, q)
    assert all(p.grad.data_ptr() >= b.flat.data_ptr() for b in tr2.generator_reducer.buckets for p in b.params)
    # reference layouts
    ref = {"generator": {"module." + k: v for k, v in ck["generator"].items()},
           "generator_ema": {"module." + k: v for k, v in ck["generator_ema"].items()},
           "discriminator": {"discriminator.module." + k: v for k, v in ck["discriminator"].items()},
           "generator_optimizer": ck["generator_optimizer"], "discriminator_optimizer": ck["discriminator_optimizer"],
           "path_length_regularization": {}}
    _, g3, d3, tr3 = _cpu_trainer(golden)
    tr3.path_length_regularization.mean_path_length = torch.tensor([0.25])
    tr3.load_checkpoint(ref)
    assert float(tr3.path_length_regularization.mean_path_length) == 0.25
    for n, q in d3.state_dict().items():
        assert torch.equal(ck["discriminator"][n], q), n
    for n, q in tr3.generator_ema.state_dict().items():
        assert torch.equal(ck["generator_ema"][n], q), n
    wrapped = tr.checkpoint_dict(data_parallel_prefix=True)
    assert all(k.startswith("module.") for k in wrapped["generator_ema"])


def test_train_loop_schedules(golden, tmp_path):
    """train(): top-k marks from the epoch / dataset length (model_wrapper.py:115-125), the wrong-order switch at
    3/4 of the epochs, checkpoints every n epochs."""
    z, g, d, tr = _cpu_trainer(golden)
    seen = []
    tr.train_iteration = lambda real, draws=None, resume_training=False, top_k=None: seen.append(
        (tr.epoch, tr.epoch >= tr.hyperparameters["wrong_order_start"] * tr.epochs, resume_training,
         None if top_k is None else (top_k.starting_iteration, top_k.final_iteration)))
    data = [torch.zeros(2, 2, 3, 32, 32)] * 3
    tr.train(data, epochs=4, save_model_after_n_epochs=2, top_k=True, checkpoint_directory=str(tmp_path))
    assert len(seen) == 12 and seen[0] == (0, False, False, (3, 9)) and seen[-1] == (3, True, False, (3, 9))
    assert sorted(os.listdir(tmp_path)) == ["checkpoint_2.pt", "checkpoint_4.pt"]
    seen.clear()
    tr.train(data, epochs=1, resume_training=True, top_k=True)
    assert seen[0][2:] == (True, (0, 1))


def _topk_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multi_stylegan_amd import dist as msg_dist
    torch.manual_seed(7)
    everything = torch.randn(world, 6)
    everything[1] -= 5.0 * (world > 1)                      # rank 1 mostly loses ...
    for v, case in ((0.5, everything), (0.5, everything * torch.tensor([[1.0], [0.0]]) + torch.tensor([[0.0], [-9.0]])),
                    (1.0, everything), (0.01, everything)):
        index, factor = msg_dist.global_top_k(case[rank], v)
        k = max(1, int(case.numel() * v))
        kept = torch.topk(case.reshape(-1), k).indices
        mine = sorted((kept[(kept >= rank * 6) & (kept < rank * 6 + 6)] - rank * 6).tolist())
        if mine:
            assert sorted(index.tolist()) == mine and abs(factor - len(mine) * world / k) < 1e-12
        else:
            assert index.tolist() == [0] and factor == 0.0    # ... and sometimes keeps nothing at all
        # the ranks' averaged weighted means equal the mean over the global k best
        local = case[rank][index].mean() * factor
        total = local.clone()
        dist.all_reduce(total)
        assert abs(total.item() / world - case.reshape(-1)[kept].mean().item()) < 1e-6
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_global_top_k_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 300
    procs = [ctx.Process(target=_topk_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"


def test_ada_controller_on_cpu_matches_oracle():
    """The device-side p controller of AdaptiveDiscriminatorAugmentation is plain tensor arithmetic: on CPU tensors it
    must follow the reference's host-side controller (oracle/ada.py, adaptive_discriminator_augmentation.py:76-94)."""
    from multi_stylegan_amd.adaptive_discriminator_augmentation import AdaptiveDiscriminatorAugmentation
    from oracle import ada as oa
    torch.manual_seed(4)
    ada = AdaptiveDiscriminatorAugmentation(torch.nn.Identity(), p_step=0.02, r_update=3, p_max=0.1)
    ref = oa.Controller(p_step=0.02, r_update=3, p_max=0.1)
    for i in range(45):
        bias = 1.0 if i < 25 else -1.0
        a, b = torch.randn(4, 1) + bias, torch.randn(4, 1, 1, 6, 6) + bias
        ada._observe(a, b)
        ref.observe(a, b, is_real=False)
        assert abs(ada.p - ref.p) < 1e-6, i
    assert len(ada.r_history) == len(ref.r_history) == 15


def test_metric_statistics_match_the_reference_values(golden):
    """multi_stylegan_amd.validation_metrics: the Frechet distance (streamed moments + symmetric eigendecomposition) against
    values the reference's own FID._calc_fid / FVD._calc_fvd produced (tests/golden/metrics.npz), whole and in ragged
    batches with a sample limit; the inception score against the oracle; the range normalisations against the reference's."""
    import numpy as np
    from multi_stylegan_amd import misc, validation_metrics as vm
    from oracle import metrics as omet
    z = golden("metrics")
    for case in ("wide", "few_samples", "shifted"):
        real, fake, want = z[f"frechet.{case}.real"], z[f"frechet.{case}.fake"], float(z[f"frechet.{case}.value"])
        got = vm.frechet_distance(real, fake)
        assert abs(got - want) <= 1e-6 * abs(want), (case, got, want)
        # streamed: ragged batches, float32 features, a limit that cuts the last batch (the reference truncates its lists)
        n = real.shape[0] - 5
        mr, mf = vm.FeatureMoments(limit=n), vm.FeatureMoments(limit=n)
        for lo in range(0, real.shape[0], 37):
            mr.update(real[lo:lo + 37].float()); mf.update(fake[lo:lo + 37].float())
        assert mr.n == mf.n == n and mr.full
        want_cut = omet.frechet_distance(real[:n].float().double().numpy(), fake[:n].float().double().numpy())
        assert abs(vm.frechet_distance_from_moments(mr, mf) - want_cut) <= 1e-6 * abs(want_cut)
    rng = np.random.default_rng(3)
    p = rng.dirichlet(np.ones(11), size=64)
    assert abs(vm.inception_score(torch.from_numpy(p)) - omet.inception_score(p)) < 1e-10
    assert torch.equal(misc.normalize_0_1_batch(z["normalize.x"]), z["normalize.y01"])
    assert torch.equal(misc.normalize_m1_1_batch(z["normalize.x"]), z["normalize.ym11"])
    frames = vm.select_frames(z["normalize.x"], 1, generator=torch.Generator().manual_seed(0))
    t = int(torch.randint(0, 2, (1,), generator=torch.Generator().manual_seed(0)))
    assert frames.shape == (3, 3, 1, 5, 4) and torch.equal(frames, omet.select_frames(z["normalize.x"], 1, t))
    with pytest.raises(ValueError, match="feature network"):
        vm.FID(None)


class _ToyGenerator(torch.nn.Module):
    """Stands in for the generator in the statistics sweeps (any module with `latent_dimensions` mapping latents to
    [B, 2, 3, H, W] images will do; the real one needs the GPU)."""
    latent_dimensions = 6

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 2 * 3 * 8 * 8)

    def forward(self, input):
        z = input[0] if isinstance(input, (list, tuple)) else input
        return torch.sigmoid(self.lin(z)).view(-1, 2, 3, 8, 8)


class _ToyFeatures(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.rows = []

    def forward(self, x):
        f = torch.stack([x.mean(dim=tuple(range(1, x.ndim))), x.flatten(1).std(dim=1), x.flatten(1)[:, 0],
                         x.flatten(1)[:, -1] * 2.0], dim=1)
        self.rows.append(f.double())
        return f


def _validation_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from multi_stylegan_amd import validation_metrics as vm
    from oracle import metrics as omet
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    gen, net = _ToyGenerator(), _ToyFeatures()
    for p_ in gen.parameters():
        dist.broadcast(p_.data, 0)
    dataset = [torch.rand(3, 2, 3, 8, 8) for _ in range(5)]                  # rank-distinct shard: 15 rows
    metric = vm.FID(net, device="cpu", batch_size=3, data_samples=20, no_rfp=True, no_gfp=True)
    got = metric(gen, dataset)
    # every rank swept ceil(20 / 2) = 10 rows of its data (4 batches, the last cut) and 10 generated rows (4 batches)
    assert len(net.rows) == 8
    mine = torch.stack([torch.cat(net.rows[:4])[:10], torch.cat(net.rows[4:])[:10]])
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    real, fake = torch.cat([b[0] for b in both]), torch.cat([b[1] for b in both])
    want = omet.frechet_distance(real.numpy(), fake.numpy())
    assert abs(got - want) <= 1e-9 * max(abs(want), 1.0), (got, want)
    assert metric.moments_real[0].n == 20
    scores = [torch.tensor([got], dtype=torch.float64) for _ in range(world)]
    dist.all_gather(scores, torch.tensor([got], dtype=torch.float64))
    assert scores[0].item() == scores[1].item()                            # the same score on every rank
    # inception score: every rank's 10 probability rows gathered
    net2 = _ToyFeatures()
    is_got = vm.IS(net2, device="cpu", batch_size=5, data_samples=20, no_rfp=True, no_gfp=True, input_size=None)(gen)
    probs = torch.cat(net2.rows).float().softmax(dim=1)
    parts = [torch.empty_like(probs) for _ in range(world)]
    dist.all_gather(parts, probs)
    assert abs(is_got - omet.inception_score(torch.cat(parts).numpy())) < 1e-6
    # a dataset that runs out before data_samples rows: says so instead of caching short statistics silently
    short = vm.FID(_ToyFeatures(), device="cpu", batch_size=3, data_samples=40, no_rfp=True, no_gfp=True)
    with pytest.warns(UserWarning, match="fewer than data_samples"):
        short(gen, dataset)
    assert short.moments_real[0].n == 30
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_validation_statistics_are_shared_over_ranks_gloo_world2():
    """A data-parallel validation pass: each rank sweeps its share of the samples, the additive moments (and the inception
    score's probability rows) are exchanged, every rank reports the statistic of the union (advisor, round 4)."""
    _run_ranks(_validation_worker, 2, 29100 + os.getpid() % 300, 120)


def test_generated_call_wrappers_bind_every_entry_point():
    """multi_stylegan_amd._msg_fastcall (generated by csrc_host/gen_fastcall.py, built by build()): one wrapper per entry point
    that returns a status / size, bound to the dlopen handle ctypes holds; same results as ctypes on the entry points that
    need no GPU, argument-count and range errors raised like ctypes raises them."""
    from multi_stylegan_amd import _lib
    h = _lib.lib()
    assert h.fastcall, "build() did not produce _msg_fastcall.so (or MSG_NO_FASTCALL is set)"
    from multi_stylegan_amd import _msg_fastcall as fast
    ints = [n for n, (res, _a) in _lib._SIGNATURES.items() if res in (_lib._I, _lib._L)]
    assert all(callable(getattr(fast, n)) for n in ints)
    assert all(getattr(h, n) is getattr(fast, n) for n in ints)            # lib() hands out the wrappers ...
    assert h.msg_build_arch() == b"gfx950"                                # ... and ctypes for the string-valued entries
    assert h.msg_abi_version() == h._ctypes.msg_abi_version() == _lib.ABI_VERSION
    geoms = [(1, 16, 256, 256, 512, 512, 256, 256, 512, 3, 3, 0), (1, 32, 256, 256, 128, 128, 256, 256, 128, 3, 3, 0),
             (1, 16, 4, 4, 512, 512, 4, 4, 512, 3, 3, 512 * 9 * 512), (0, 2, 9, 9, 16, 32, 9, 9, 24, 3, 3, 0)]
    for g in geoms:
        assert h.msg_conv2d_fprop_plan(*g) == h._ctypes.msg_conv2d_fprop_plan(*g)
    assert h.msg_bias_act_backward_workspace(1 << 20, 1, 512, 1) == h._ctypes.msg_bias_act_backward_workspace(1 << 20, 1, 512, 1)
    assert h.msg_softmax_rows(None, None, 0, 4, 1024, None) == -1         # MSG_EINVAL: None is a NULL pointer, as with ctypes
    with pytest.raises(TypeError):
        h.msg_conv2d_fprop_plan(1, 2, 3)
    with pytest.raises(OverflowError):
        h.msg_conv2d_fprop_plan(1 << 40, *geoms[0][1:])
    with pytest.raises(TypeError):
        h.msg_conv2d_fprop_plan("1", *geoms[0][1:])


def test_bench_multi_rank_fields():
    """What bench.py's JSON line reports about the ranks (the fields the 8-GPU run is read by), on canned timings."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.multi_rank_fields(4, "nccl", False, 16, 20, [2.0, 2.5, 2.2, 2.4], 131.0)
    assert f["rccl_ranks"] == 4 and f["backend"] == "nccl" and f["rehearsal_shared_gpu"] is False
    assert f["per_rank_img_per_s"] == [160.0, 128.0, 145.45, 133.33]
    assert f["overlap"] == {"on_ms_per_step": 125.0, "off_ms_per_step": 131.0}        # the slowest rank's 2.5 s / 20 steps
    one = bench.multi_rank_fields(1, None, False, 16, 20, [2.25], None)
    assert one["rccl_ranks"] == 1 and one["backend"] is None and one["overlap"] is None
    reh = bench.multi_rank_fields(2, "gloo", True, 4, 3, [1.0, 1.0], 340.0)
    assert reh["backend"] == "gloo" and reh["rehearsal_shared_gpu"] is True
