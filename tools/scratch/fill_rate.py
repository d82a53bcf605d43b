import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools.microbench import timeit
dev="cuda:0"
out = torch.empty(16, 512, 9, 512, device=dev, dtype=torch.bfloat16)
src = torch.randn(16, 512, 9, 512, device=dev).to(torch.bfloat16)
t = timeit(lambda: out.zero_(), 20); print(f"zero_ 75MB: {t*1e6:.1f} us {out.numel()*2/t/1e9:.0f} GB/s")
t = timeit(lambda: out.copy_(src), 20); print(f"copy_ 75MB: {t*1e6:.1f} us {2*out.numel()*2/t/1e9:.0f} GB/s (r+w)")
from multi_stylegan_amd import conv_ops, _lib
base = torch.randn(512, 9, 512, device=dev); wsq = torch.rand(512, 512, device=dev); s = torch.rand(16, 512, device=dev)
d = torch.empty(16, 512, device=dev)
def mw():
    _lib.check(_lib.lib().msg_modulate_weights(base.data_ptr(), wsq.data_ptr(), s.data_ptr(), out.data_ptr(), d.data_ptr(), 1, 16, 512, 512, 9, 512, 512, 0.1, 1e-8, _lib.stream_of(torch.device(dev))), "mw")
t = timeit(mw, 20); print(f"modulate_weights: {t*1e6:.1f} us {(out.numel()*2 + base.numel()*4)/t/1e9:.0f} GB/s")
