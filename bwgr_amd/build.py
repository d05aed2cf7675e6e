"""Build libbwgr_hip.so in-tree for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BWGR_LIB: an already built library to load instead -- tools/variants.py compares compile-time variants of the kernels that way; never built here)
LIB = os.environ.get("BWGR_LIB") or os.path.join(_HERE, "libbwgr_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("bwgr_hip.hip",)]
# every file under csrc/ (the kernels live in headers that bwgr_hip.hip includes) plus the public header
DEPS = sorted(glob.glob(os.path.join(_HERE, "csrc", "*"))) + [os.path.join(_HERE, "..", "include", "bwgr.h")]
# -ffp-contract=off: scalar float arithmetic must round exactly where the reference's does;
# fused multiply-adds are written explicitly (fma / __fmul_rn / __fsub_rn) where intended.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def needs_build():
    if os.environ.get("BWGR_LIB"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
