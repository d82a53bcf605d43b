import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multi_stylegan_amd as m
from multi_stylegan_amd.op_static import fused_act
from conftest import Golden
from test_hip_models import _golden_trainer
from test_oracle_golden import load_train_draws
cache = {}
def golden(n):
    if n not in cache: cache[n] = Golden(n)
    return cache[n]
orig = fused_act.FusedLeakyReLUFunctionBackward.forward
calls = []
def fwd(ctx, grad_output, out, noise, need_bias, negative_slope, scale):
    res = orig(ctx, grad_output, out, noise, need_bias, negative_slope, scale)
    calls[-1].append((grad_output.detach().clone(), out.detach().clone(), res[1].detach().clone(), grad_output.shape, grad_output.stride(), out.stride(),
                      grad_output.data_ptr() % 256, out.data_ptr() % 256, scale))
    return res
fused_act.FusedLeakyReLUFunctionBackward.forward = staticmethod(fwd)
for run in range(2):
    calls.append([])
    z, g, d, trainer = _golden_trainer(golden)
    real, draws = load_train_draws(z, 1, m.model_wrapper)
    trainer.iteration = 15
    trainer.train_iteration(real.to("cuda:0"), draws.to("cuda:0"))
    torch.cuda.synchronize()
a, b = calls
print(len(a), len(b))
for i, (ca, cb) in enumerate(zip(a, b)):
    same_in = torch.equal(ca[0], cb[0]) and torch.equal(ca[1], cb[1])
    same_out = torch.equal(ca[2], cb[2])
    if not same_out or not same_in:
        print(i, "in_same", same_in, "gb_same", same_out, ca[3], ca[4], ca[5], ca[6:], cb[4], cb[5], cb[6:])
        # re-run the kernel on the captured inputs several times
        outs = [fused_act.FusedLeakyReLUFunctionBackward.apply(ca[0], ca[1], None, True, 0.2, ca[8])[1] for _ in range(4)]
        print("   rerun equal:", [torch.equal(outs[0], o) for o in outs], "vs a", torch.equal(outs[0], ca[2]), "vs b", torch.equal(outs[0], cb[2]))
