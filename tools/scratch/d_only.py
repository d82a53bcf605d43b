import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
D = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True).to(dev)
D.compute_dtype = torch.bfloat16
real = torch.rand(16, 2, 3, 256, 256, device=dev)
for _ in range(13):
    a, b = D(real, is_real=True, is_cut_mix=False); (a.mean() + b.mean()).backward()
torch.cuda.synchronize()
