import math, sys, torch
sys.path.insert(0, '.')
from multi_stylegan_amd import conv_ops
DEV='cuda:0'
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
for (B,i,o,hw,ps) in [(16,512,6,256,True),(32,128,1,256,False),(16,128,6,256,False),(16,6,512,256,True),(32,6,128,256,False),(32,1,128,256,False),(16,512,6,128,True)]:
    x = conv_ops.to_compute_layout(torch.randn(B,i,hw,hw,device=DEV), torch.bfloat16)
    w = torch.randn(*((B,o,i,1,1) if ps else (o,i,1,1)),device=DEV)/math.sqrt(i)
    geo = conv_ops.Geometry("conv",1,1,1,0,(hw,hw),ps)
    y = conv_ops._f_raw(x,w,None,geo)
    t = timeit(lambda: conv_ops._f_raw(x,w,None,geo))
    xv,cx = conv_ops._nhwc_view(x)
    by = (B*hw*hw*cx + B*hw*hw*y.stride(3))*2
    print(f"B{B} {i}->{o} @{hw} ps={ps}: {t:8.1f} us  {by/t/1e3:7.1f} GB/s", flush=True)
y = torch.empty(16, 256, 256, 512, device=DEV, dtype=torch.bfloat16)
print(f"fill 1 GiB: {timeit(lambda: y.zero_()):.1f} us -> {y.numel()*2/timeit(lambda: y.zero_())/1e3:.1f} GB/s")
z = torch.empty_like(y)
print(f"copy 1 GiB: {y.numel()*4/timeit(lambda: z.copy_(y))/1e3:.1f} GB/s (read + write)")
