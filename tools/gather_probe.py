import sys, torch
sys.path.insert(0,'.')
from multi_stylegan_amd import conv_ops
DEV='cuda:0'
x = conv_ops.to_compute_layout(torch.randn(32,6,256,256,device=DEV), torch.bfloat16)
geo = conv_ops.Geometry("conv",3,3,1,1,(256,256),False)
f=lambda: conv_ops._gather_taps(x,6,geo)
for _ in range(3): f()
torch.cuda.synchronize()
a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
t=a.elapsed_time(b)/20*1e3
print(f"gather_taps B32 6ch 256^2: {t:.1f} us, {(32*65536*(16+128))/t/1e3:.0f} GB/s")
