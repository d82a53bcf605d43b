import os, sys, random, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import multi_stylegan_amd as m
from multi_stylegan_amd.config import generator_config_for_resolution
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
gen = m.MultiStyleGANGenerator(generator_config_for_resolution(256))
dis = m.MultiStyleGANDiscriminator(m.u_net_2d_discriminator_config, no_rfp=True)
gen.compute_dtype = dis.compute_dtype = torch.bfloat16
trainer = m.ModelWrapper(gen, dis, device=dev)
real = torch.rand(16, 2, 3, 256, 256, device=dev)
for _ in range(2): trainer.train_iteration(real)
torch.cuda.synchronize()
counts = collections.Counter()
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "multi_stylegan_amd" in fr.filename or "bench" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "torch-internal"
import functools
def wrap(mod, name):
    orig = getattr(mod, name)
    @functools.wraps(orig)
    def f(*a, **k):
        counts[(name, where())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n)
for n in ("zero_", "fill_", "contiguous", "to", "float", "clone", "copy_"):
    wrap(torch.Tensor, n)
trainer.train_iteration(real)
torch.cuda.synchronize()
for (name, w), c in counts.most_common(45):
    print(f"{c:5d}  {name:12s} {w}")
