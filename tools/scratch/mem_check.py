import os, sys, random, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
tr = m.ModelWrapper(gen, dis, device=dev)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
for i in range(1, 49):
    tr.train_iteration(real)
    if i % 8 == 0:
        torch.cuda.synchronize()
        logs = tr.pop_logs()
        print(i, f"alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated()/2**30:.2f}",
              {k: round(v[-1], 3) for k, v in logs.items() if "loss_generator" == k or k == "path_length"}, flush=True)
