import json, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import conftest
from conftest import Golden
import test_hip_models as T
cache = {}
def golden(name):
    if name not in cache: cache[name] = Golden(name)
    return cache[name]
loose = {label: (1.0, 1.0, 10.0) for label in ("d", "g", "cm_aug", "cm_reg", "r1", "pl")}
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "noderive":
    import multi_stylegan_amd.conv_ops as co, multi_stylegan_amd.op_static.fused_act as fa, multi_stylegan_amd.op_static.upfirdn2d as up
    import multi_stylegan_amd.op_static.rgb_skip as rs, multi_stylegan_amd.op_static.maxpool as mp, multi_stylegan_amd.op_static.softmax as sm
    for mod in (co, fa, up, rs, mp, sm):
        mod._derive = lambda fn, *a: fn.apply(*a)
for fused in ("plain", "flat"):
    try:
        rep = T._run_golden_iterations(golden, fused, step_tol=loose)
        print(mode, fused, json.dumps({k: ({kk: (round(vv, 7) if isinstance(vv, float) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in rep.items()}))
    except AssertionError as e:
        print(mode, fused, "ASSERT", str(e)[:500])
