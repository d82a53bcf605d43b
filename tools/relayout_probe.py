import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools.microbench import timeit
from multi_stylegan_amd import _lib
DEV="cuda:0"
for (o,i,t) in ((512,512,9),(512,512,4),(128,128,9),(256,128,1)):
    w=torch.randn(o,i,t,device=DEV)
    fwd=torch.empty(o,t,i,device=DEV,dtype=torch.bfloat16); dg=torch.empty(i,t,o,device=DEV,dtype=torch.bfloat16); wsq=torch.empty(o,i,device=DEV)
    st=_lib.stream_of(torch.device(DEV))
    def fn():
        c=_lib.lib().msg_relayout_weight(w.data_ptr(), fwd.data_ptr(), dg.data_ptr(), wsq.data_ptr(), _lib.MSG_BF16, o,i,t,i,o,1,0,0.5,st); assert c==0
    tt=timeit(fn,200,warm=20)
    print(f"relayout O{o} I{i} T{t}: {tt*1e6:.1f} us", flush=True)
    # check against torch
    fn(); torch.cuda.synchronize()
    want_f=(w*0.5).permute(0,2,1).bfloat16(); want_d=(w*0.5).flip(2).permute(1,2,0).bfloat16()
    assert torch.equal(fwd, want_f.contiguous()) and torch.equal(dg, want_d.contiguous()), "mismatch"
    assert torch.allclose(wsq, w.square().sum(2), rtol=1e-5)
print("ok")
