"""Build the gfx950 shared library (zvec_amd/libzvec_hip.so) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "zvec_hip_api.hip")
DEPS = [SRC, os.path.join(os.path.dirname(HERE), "include", "zvec_hip.h")] + sorted(
    os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h"))
OUT = os.path.join(HERE, "libzvec_hip.so")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return OUT
    hipcc = _hipcc()
    if hipcc is None:
        if os.path.exists(OUT):
            return OUT  # GPU box without a compiler on PATH: use the prebuilt library
        raise RuntimeError("hipcc not found and %s is missing" % OUT)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
