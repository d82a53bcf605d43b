import os, sys, time, random, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
trainer.generator_ema.compute_dtype = torch.bfloat16
random.seed(1)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
for i in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.train_iteration(real)
    torch.cuda.synchronize()
    print(i + 1, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
