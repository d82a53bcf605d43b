import json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
import torch
import conftest
from conftest import Golden
orig = conftest.check_step_trace
HIST = {}
def spy(got, ref, tol_grad, tol_norm, tol_delta, lr_floor=0.05, history=None):
    names = sorted(k[len("grad."):] for k in ref if k.startswith("grad."))
    for n in names:
        g_ref, g_got = ref["grad." + n].double().cpu(), got["grad." + n].double().cpu()
        d_ref, d_got = ref["delta." + n].double().cpu(), got["delta." + n].double().cpu()
        gmax, dmax = g_ref.abs().max().item(), d_ref.abs().max().item()
        if gmax == 0 or dmax == 0:
            HIST.setdefault(n, []).append((g_ref, g_got)); continue
        mask = g_ref.abs() > lr_floor * gmax
        err = ((d_got - d_ref).abs() * mask) / dmax
        e = err.max().item()
        if e > 2e-3:
            idx = int(err.argmax())
            f = lambda t: t.flatten()[idx].item()
            print(f"STEP worst element of {n}: idx {idx} err {e:.3e}  g_ref {f(g_ref):.6e} g_got {f(g_got):.6e} gmax {gmax:.3e}  d_ref {f(d_ref):.6e} d_got {f(d_got):.6e} dmax {dmax:.3e}")
            for k, (hr, hg) in enumerate(HIST.get(n, [])):
                print(f"     earlier step {k}: g_ref {f(hr):.6e} g_got {f(hg):.6e}  max {hr.abs().max().item():.3e}")
        HIST.setdefault(n, []).append((g_ref, g_got))
    return orig(got, ref, 1.0, 1.0, 10.0, lr_floor, history)
conftest.check_step_trace = spy
import test_hip_models as T
T.check_step_trace = spy
cache = {}
def golden(name):
    if name not in cache: cache[name] = Golden(name)
    return cache[name]
for fused in ("plain",):
    HIST.clear()
    rep = T._run_golden_iterations(golden, fused)
    print(fused, {k: (v["worst_delta"] if isinstance(v, dict) else v) for k, v in rep.items()})
