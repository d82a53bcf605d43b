import os, sys, torch, torch.multiprocessing as mp
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_ddp as t
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    res = []
    for i in range(2):
        q = ctx.Queue(); p = ctx.Process(target=t._plain_worker, args=(q,)); p.start(); res.append(q.get(timeout=600)); p.join()
    q = ctx.Queue(); p = ctx.Process(target=t._rccl_worker, args=(29950, q)); p.start(); res.append(q.get(timeout=600)); p.join()
    for name, (a, b) in {"plain vs plain": (res[0], res[1]), "plain vs rccl": (res[0], res[2])}.items():
        rel = (a[0] - b[0]).norm() / a[0].norm()
        print(name, "param rel", float(rel))
        for k in a[1]:
            print("   ", k, a[1][k], b[1][k])
