"""Data-parallel trainer on the real GPU: two ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device;
gloo accepts GPU tensors), so the hook -> side-stream -> async all-reduce -> finish() path of GradBucketReducer runs
with real HIP streams.  Checks that the replicas stay bit-identical and -- through tests/ddp_probe.py, which snapshots
every bucket's local gradient as it is handed to the collective -- that each optimiser step equals the step a single
process would take with the mean of the shards' gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out, sgd):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multi_stylegan_amd as m
    from ddp_probe import StepProbe
    from tools.gen_golden import TINY_D, TINY_G
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(10 + rank)                                   # different init per rank: broadcast must fix it
    g, d = m.MultiStyleGANGenerator(TINY_G), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    # sgd=True: plain SGD through _step's clip_ branch (movement proportional to the exchanged gradient);
    # sgd=False: the product default, fused Adam with 1/world and the clip folded into grad_scale
    opts = dict(generator_optimizer=torch.optim.SGD(g.parameters(), lr=1e-3),
                discriminator_optimizer=torch.optim.SGD(d.parameters(), lr=1e-3)) if sgd else {}
    tr = m.ModelWrapper(g, d, device="cuda:0", bucket_bytes=1 << 15, **opts)
    assert tr.generator_reducer.comm_stream is not None and len(tr.generator_reducer.buckets) > 2
    probe = StepProbe(tr)
    tr.iteration = 15                                              # -> iteration 16: R1 and path length fire too
    torch.manual_seed(1000 + rank)
    for it in range(2):
        tr.step_trace.clear(); probe.local.clear()
        tr.train_iteration(torch.rand(2, 2, 3, 32, 32, device="cuda:0"))
        # every optimiser step == the single-process step with the mean of the two shards' gradients
        assert probe.check(world, lr=1e-3 if sgd else None) == (["d", "g", "pl", "r1"] if it == 0 else ["d", "g"])
    logs = tr.pop_logs()
    assert all(all(v == v for v in vals) for vals in logs.values()), "NaN in losses"
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("sgd", [True, False])
def test_trainer_two_ranks_one_gpu_gloo(sgd):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (7 if sgd else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sgd)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"


def _rccl_worker(port, out):
    """ONE rank, real RCCL communicator, collectives forced on: must reproduce the plain single-process run bit for bit
    (a one-rank all-reduce / broadcast is the identity)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MSG_FORCE_COLLECTIVES="1")
    sys.path.insert(0, ROOT)
    import multi_stylegan_amd as m
    from multi_stylegan_amd import dist as msg_dist
    from tools.gen_golden import TINY_D, TINY_G
    msg_dist.init_from_env()                                       # backend "nccl" == RCCL
    assert dist.is_initialized() and dist.get_backend() == "nccl" and msg_dist.collectives_active()
    torch.manual_seed(10)
    g, d = m.MultiStyleGANGenerator(TINY_G).to("cuda:0"), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True).to("cuda:0")
    # plain SGD: a gradient perturbation moves the parameters proportionally (Adam with beta1 = 0 would turn the
    # float-atomic noise of the weight-gradient kernels into steps of size lr and hide a 1e-3 exchange error)
    tr = m.ModelWrapper(g, d, device="cuda:0", bucket_bytes=1 << 15,
                        generator_optimizer=torch.optim.SGD(g.parameters(), lr=1e-3),
                        discriminator_optimizer=torch.optim.SGD(d.parameters(), lr=1e-3))
    assert tr.generator_reducer.active and tr.generator_reducer.comm_stream is not None
    tr.iteration = 15
    torch.manual_seed(1000)
    import random
    random.seed(1000)                                              # style-mixing draws come from Python's RNG ...
    import numpy
    numpy.random.seed(1000)                                        # ... and the crossover layer from numpy's (as in the reference)
    for _ in range(2):
        tr.train_iteration(torch.rand(2, 2, 3, 32, 32, device="cuda:0"))
    logs = tr.pop_logs()
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())]).cpu()
    out.put((flat, logs))
    dist.destroy_process_group()


def _plain_worker(out):
    sys.path.insert(0, ROOT)
    import multi_stylegan_amd as m
    from tools.gen_golden import TINY_D, TINY_G
    torch.manual_seed(10)
    g, d = m.MultiStyleGANGenerator(TINY_G).to("cuda:0"), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True).to("cuda:0")
    tr = m.ModelWrapper(g, d, device="cuda:0", bucket_bytes=1 << 15,
                        generator_optimizer=torch.optim.SGD(g.parameters(), lr=1e-3),
                        discriminator_optimizer=torch.optim.SGD(d.parameters(), lr=1e-3))
    tr.iteration = 15
    torch.manual_seed(1000)
    import random
    random.seed(1000)                                              # style-mixing draws come from Python's RNG ...
    import numpy
    numpy.random.seed(1000)                                        # ... and the crossover layer from numpy's (as in the reference)
    for _ in range(2):
        tr.train_iteration(torch.rand(2, 2, 3, 32, 32, device="cuda:0"))
    logs = tr.pop_logs()
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())]).cpu()
    out.put((flat, logs))


def test_trainer_one_rank_rccl_collectives_forced():
    ctx = mp.get_context("spawn")
    q1, q2 = ctx.Queue(), ctx.Queue()
    p1 = ctx.Process(target=_rccl_worker, args=(29900 + os.getpid() % 90, q1))
    p1.start()
    flat1, logs1 = q1.get(timeout=600)
    p1.join(120)
    p2 = ctx.Process(target=_plain_worker, args=(q2,))
    p2.start()
    flat2, logs2 = q2.get(timeout=600)
    p2.join(120)
    assert p1.exitcode == 0 and p2.exitcode == 0
    assert all(all(v == v for v in vals) for vals in logs1.values()), "NaN in losses"
    # same trajectory (every kernel is deterministic; the all-reduce of one rank is the identity); a broken exchange (double counting,
    # a missing bucket, a stale gradient) changes the SGD step at the 1e-1 level
    rel = (flat1 - flat2).norm() / flat2.norm()
    assert rel < 1e-5, rel
    for key in logs2:
        a, b = torch.tensor(logs1[key]), torch.tensor(logs2[key])
        assert torch.allclose(a, b, rtol=2e-3, atol=1e-4), (key, logs1[key], logs2[key])


def _rccl_two_rank_worker(rank, world, port, out, exchange):
    """TWO ranks on two GPUs over RCCL: the launch the 8-GPU benchmark makes, at test size."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MSG_DDP_EXCHANGE=exchange)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import copy
    import multi_stylegan_amd as m
    from multi_stylegan_amd import dist as msg_dist
    from ddp_probe import StepProbe
    from tools.gen_golden import TINY_D, TINY_G
    _, _, local_rank = msg_dist.init_from_env()
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    dev = f"cuda:{local_rank}"
    torch.manual_seed(10 + rank)                                   # different init per rank: the broadcast must fix it
    g0, d0 = m.MultiStyleGANGenerator(TINY_G), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    finals = []
    for overlap in (True, False):
        g, d = copy.deepcopy(g0), copy.deepcopy(d0)
        tr = m.ModelWrapper(g, d, device=dev, bucket_bytes=1 << 15, overlap_communication=overlap)
        assert tr.generator_reducer.active and tr.generator_reducer.exchange == exchange
        probe = StepProbe(tr)
        tr.iteration = 14
        torch.manual_seed(1000 + rank)
        import random
        import numpy
        random.seed(1000 + rank); numpy.random.seed(1000 + rank)
        for it in range(3):                                        # 15 (plans are learned), 16 (R1 + path length), 17 (plans used)
            tr.step_trace.clear(); probe.local.clear()
            tr.train_iteration(torch.rand(2, 2, 3, 32, 32, device=dev))
            assert probe.check(world) == (["d", "g", "pl", "r1"] if it == 1 else ["d", "g"])
        logs = tr.pop_logs()
        assert all(all(v == v for v in vals) for vals in logs.values()), "NaN in losses"
        flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), "replicas diverged"
        finals.append(flat.cpu())
    # overlapping the exchange with backward changes WHEN buckets travel, not what arrives
    assert torch.equal(finals[0], finals[1]), "overlap on / off disagree"
    if rank == 0:
        out.put("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all_reduce", "reduce_scatter"])
def test_trainer_two_ranks_two_gpus_rccl(exchange):
    """Runs wherever two GPUs are visible (the driver's 8-GPU node; skipped on the 1-GPU test boxes): two RCCL ranks, every
    optimiser step == the step of the mean of the shards' gradients (tests/ddp_probe.py), replicas bit-identical, and the
    overlapped exchange equal to the exchange after backward -- so that the first multi-rank RCCL launch is not the benchmark."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 200 + (11 if exchange == "reduce_scatter" else 0)
    procs = [ctx.Process(target=_rccl_two_rank_worker, args=(r, 2, port, q, exchange)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(600) for p in procs]
    hung = [p for p in procs if p.exitcode is None]
    [p.kill() for p in hung]
    assert not hung and all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"


def test_arm_label_refuses_a_dirty_store():
    """The direct route WRITES gradients into the flat store, so a labelled backward must start from zero_grad() -- whichever
    way the store was dirtied: a labelled backward, an unlabelled arm() + backward, or plain AccumulateGrad while the reducer
    was not armed at all (advisor, round 4: only the first of the three used to be tracked)."""
    sys.path.insert(0, ROOT)
    from multi_stylegan_amd.dist import GradBucketReducer
    torch.manual_seed(0)
    lin = torch.nn.Linear(8, 8).to("cuda:0")
    red = GradBucketReducer(lin.parameters(), bucket_bytes=1 << 10)
    x = torch.randn(4, 8, device="cuda:0")

    def backward(label, arm=True):
        if arm:
            red.arm(label)
        lin(x).sum().backward()
        if arm:
            red.finish()

    red.zero_grad(); backward("a")                                 # the ordinary sequence
    want = lin.weight.grad.clone()
    with pytest.raises(RuntimeError, match="zero_grad"):
        red.arm("a")                                               # labelled after labelled
    red.disarm()
    red.zero_grad(); backward(None)                                # unlabelled arm + backward dirties it
    assert torch.equal(lin.weight.grad, want)
    with pytest.raises(RuntimeError, match="zero_grad"):
        red.arm("a")
    red.disarm()
    red.zero_grad(); backward(None, arm=False)                     # not armed at all: AccumulateGrad into the attached views
    assert torch.equal(lin.weight.grad, want)
    with pytest.raises(RuntimeError, match="zero_grad"):
        red.arm("a")
    red.disarm()
    red.zero_grad(); backward("a")
    assert torch.equal(lin.weight.grad, want)


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` end to end as the driver starts it -- self-launch through torch.distributed.run from a parent that
    never touches the GPU, rendezvous on 127.0.0.1, the bucket exchange during backward, the overlap-off leg, rank 0's JSON
    line -- on two ranks sharing this box's one GPU over gloo (RCCL refuses two ranks on one device): the first multi-rank run
    of the benchmark must not be the 8-GPU measurement itself (reference: train_multi_stylegan.py:67-70)."""
    import json
    import subprocess
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MSG_LIB_VARIANT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-fp32-leg"]
    t0 = time.time()
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    took = time.time() - t0
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]                       # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo" and out["rehearsal_shared_gpu"] is True
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert len(out["per_rank_img_per_s"]) == 2 and all(v > 0 for v in out["per_rank_img_per_s"])
    assert out["overlap"]["on_ms_per_step"] > 0 and out["overlap"]["off_ms_per_step"] > 0
    assert out["value"] > 0 and out["library"]["variant"] is None
    assert out["losses"] and all(v == v and abs(v) != float("inf") for v in out["losses"].values())
    print(f"two-rank rehearsal: {took:.0f} s, {out['value']} img/s on one shared GPU, overlap {out['overlap']}")
    assert took < 120, took
