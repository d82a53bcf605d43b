import os, sys, time, random, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
from multi_stylegan_amd import misc
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
G = m.MultiStyleGANGenerator(generator_config_for_resolution(256)).to(dev)
D = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True).to(dev)
G.compute_dtype = D.compute_dtype = torch.bfloat16
real = torch.rand(16, 2, 3, 256, 256, device=dev)
def noise():
    return misc.get_noise(batch_size=16, latent_dimension=512, p_mixed_noise=0.9, device=dev)
def timed(name, fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms", flush=True)
random.seed(0)
def g_nograd():
    with torch.no_grad(): return G(input=noise())
fake = g_nograd()
def d_fwd_nograd():
    with torch.no_grad(): D(real, is_real=True, is_cut_mix=False)
def d_fwd(): return D(real, is_real=True, is_cut_mix=False)
def d_fwd_bwd():
    a, b = D(real, is_real=True, is_cut_mix=False); (a.mean() + b.mean()).backward()
def g_fwd(): return G(input=noise())
def g_fwd_bwd():
    f = G(input=noise()); f.float().mean().backward()
def gd_fwd_bwd():
    f = G(input=noise()); a, b = D(f, is_real=False, is_cut_mix=False); (a.mean() + b.mean()).backward()
timed("G forward (no grad)", g_nograd)
timed("D forward (no grad)", d_fwd_nograd)
timed("D forward (grad)", d_fwd)
timed("D forward+backward", d_fwd_bwd)
timed("G forward (grad)", g_fwd)
timed("G forward+backward", g_fwd_bwd)
timed("G->D forward+backward (G step)", gd_fwd_bwd)
