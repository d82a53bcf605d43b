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


def lib_path(variant=None):
    return LIB if not variant else os.path.join(HERE, f"libmsg_hip_{variant}.so")


def is_stale(variant=None):
    lib = lib_path(variant)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "msg_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=(), variant=None):
    """``variant``: a diagnostic / A-B build kept BESIDE the product library as libmsg_hip_<variant>.so (objects in
    build_<variant>/), loaded by tools through MSG_LIB_VARIANT=<variant>; e.g. ``--variant tuning --flags=-DMSG_TUNING`` (the
    kernel-selection constants of csrc/msg_common.h read MSG_* environment variables) or ``--variant stamps
    --flags=-DMSG_ROW3_STAMPS``.  The product library (no variant) is what tests, smoke and bench.py load."""
    lib = lib_path(variant)
    if not force and not is_stale(variant):
        build_fastcall(verbose=verbose)
        return lib
    objs = []
    objdir = os.path.join(HERE, "build" if not variant else f"build_{variant}")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
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
    cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_fastcall(force=force, verbose=verbose)
    return lib


FASTCALL = os.path.join(HERE, "_msg_fastcall.so")


def build_fastcall(force=False, verbose=True):
    """The host-side call wrappers (csrc_host/gen_fastcall.py): C source generated from _lib._SIGNATURES, compiled with gcc
    against this interpreter's headers into multi_stylegan_amd/_msg_fastcall.so.  One module serves the product library and
    every variant build (it binds to whatever handle _lib loaded)."""
    import sysconfig
    gen = os.path.join(HERE, "csrc_host", "gen_fastcall.py")
    deps = [gen, os.path.join(HERE, "_lib.py")]
    if not force and os.path.exists(FASTCALL) and all(os.path.getmtime(FASTCALL) > os.path.getmtime(d) for d in deps):
        return FASTCALL
    sys.path.insert(0, os.path.dirname(HERE))
    try:
        from multi_stylegan_amd import _lib
        from multi_stylegan_amd.csrc_host.gen_fastcall import generate
    finally:
        sys.path.pop(0)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    src = os.path.join(objdir, "msg_fastcall.c")
    with open(src, "w") as f:
        f.write(generate(_lib._SIGNATURES))
    cmd = [shutil.which("gcc") or "gcc", "-O2", "-shared", "-fPIC", f"-I{sysconfig.get_paths()['include']}", src, "-o", FASTCALL, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return FASTCALL


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--variant", default=None)
    ap.add_argument("--flags", default="", help="extra hipcc flags, e.g. --flags=-DMSG_TUNING")
    a = ap.parse_args()
    print(build(force=a.force, extra_flags=tuple(a.flags.split()), variant=a.variant))
