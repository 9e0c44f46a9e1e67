"""Builds csrc/*.hip into csrc/librbr_hip.so for gfx950 with hipcc (in-tree, no torch headers).

    python review-based-recommender_amd/build.py [--force]
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "librbr_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def extra_flags():
    """RBR_DIAG=1 in the environment builds the diagnostic library (-DRBR_DIAG: in-kernel time stamps and the RBR_DEV_*
    ablation switches of the dev tools); the product build has neither."""
    return ["-DRBR_DIAG"] if os.environ.get("RBR_DIAG") == "1" else []


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "rbr_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = _hipcc()
    objs, todo = [], []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))),
                os.path.getmtime(os.path.join(HERE, "..", "include", "rbr_hip.h"))):
            todo.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *extra_flags(), "-c", src, "-o", obj])
        objs.append(obj)

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:      # the translation units are independent: a few at a time (each hipcc is itself single-threaded)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, min(6, (os.cpu_count() or 2) - 1)))) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
