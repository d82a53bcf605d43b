import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
D = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True).to(dev)
D.compute_dtype = torch.bfloat16
r16a, r16b = torch.rand(16, 2, 3, 256, 256, device=dev), torch.rand(16, 2, 3, 256, 256, device=dev)
r32 = torch.cat([r16a, r16b])
def timed(name, fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms", flush=True)
def two():
    a, b = D(r16a, is_real=True, is_cut_mix=False); c, d = D(r16b, is_real=False, is_cut_mix=False)
    (a.mean() + b.mean() + c.mean() + d.mean()).backward()
def one():
    a, b = D(r32, is_real=True, is_cut_mix=False); (a.mean() + b.mean()).backward()
timed("2 x D(B=16) fwd+bwd", two)
timed("1 x D(B=32) fwd+bwd", one)
