"""Build libmsg_hip.so (the C-ABI shared library of include/msg_hip.h) in-tree with hipcc for gfx950.

``python -m multi_stylegan_amd.build`` cross-compiles without a GPU.  The .so is git-ignored but travels to
the GPU box with the repo snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsg_hip.so")
ARCH = "gfx950"


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "msg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    if not force and not is_stale():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        deps = [src] + glob.glob(os.path.join(CSRC, "*.h"))
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps):
            continue
        cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", src, "-o", obj,
               "-Wno-unused-result", *extra_flags, *os.environ.get("MSG_EXTRA_HIPCC_FLAGS", "").split()]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
