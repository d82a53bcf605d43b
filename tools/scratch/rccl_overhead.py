"""One rank, RCCL communicator, collectives forced: cost of the gradient exchange machinery (run under torchrun or plain)."""
import os, sys, time, random, torch
os.environ.setdefault("MSG_FORCE_COLLECTIVES", "1")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
from multi_stylegan_amd import dist as msg_dist
from multi_stylegan_amd.config import generator_config_for_resolution
msg_dist.init_from_env()
dev = torch.device("cuda", 0)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
def run(overlap, bucket_mb, label):
    torch.manual_seed(0); random.seed(0)
    gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
    dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
    gen.compute_dtype = dis.compute_dtype = torch.bfloat16
    tr = m.ModelWrapper(gen, dis, device=dev, overlap_communication=overlap, bucket_bytes=bucket_mb << 20)
    for _ in range(3): tr.train_iteration(real)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.train_iteration(real)
    torch.cuda.synchronize()
    print(f"{label:40s} {(time.perf_counter() - t0) / 10 * 1e3:7.1f} ms/iter  buckets D={len(tr.discriminator_reducer.buckets)} G={len(tr.generator_reducer.buckets)}", flush=True)
    del tr, gen, dis
    torch.cuda.empty_cache()
run(True, 32, "overlap, 32 MiB buckets")
run(False, 32, "no overlap, 32 MiB buckets")
run(True, 256, "overlap, 256 MiB buckets")
run(False, 1024, "no overlap, one bucket")
