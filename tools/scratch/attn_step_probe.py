"""Why is the full step slower with the fused attention although its kernels are not?  Wall time per iteration, device
allocations (hipMalloc calls synchronise) and GPU-busy time, with MSG_FUSED_ATTENTION from the environment."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
torch.manual_seed(0)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
tr = m.ModelWrapper(gen, dis, device="cuda:0")
tr.generator_ema.compute_dtype = torch.bfloat16
real = torch.rand(16, 2, 3, 256, 256, device="cuda:0")
for _ in range(4):
    tr.train_iteration(real)
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats()
t0 = time.perf_counter()
host = []
REG = bool(int(os.environ.get("PROBE_REG", "0")))       # 1: every measured iteration is a regularised (16th) one
if REG:
    tr.iteration = 15
    tr.train_iteration(real)                            # warm the second-order paths up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
for _ in range(10):
    h0 = time.perf_counter()
    if REG:
        tr.iteration = 15
    tr.train_iteration(real)
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
s1 = torch.cuda.memory_stats()
print(f"fused={os.environ.get('MSG_FUSED_ATTENTION', '1')} wall {dt * 1e3:.1f} ms/it  host-enqueue {sum(host) / 10 * 1e3:.1f} ms/it  "
      f"device mallocs {s1['num_device_alloc'] - s0['num_device_alloc']} frees {s1['num_device_free'] - s0['num_device_free']} "
      f"retries {s1['num_alloc_retries'] - s0['num_alloc_retries']} reserved {s1['reserved_bytes.all.peak'] / 2**30:.1f} GiB")
