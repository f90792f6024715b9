"""Builds xpoly_amd/libxpoly_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

-ffp-contract=off is part of the numerical contract, not a tuning flag: the
reference rounds after the multiply and again after the add (lpsol.h:1485-1489).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "xpoly_amd.hip")
OUT = os.path.join(HERE, "libxpoly_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def sources():
    d = os.path.join(HERE, "csrc")
    inc = os.path.join(HERE, "..", "include", "xpoly_amd.h")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [inc]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False):
    if not force and not stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc] + FLAGS + ["-o", OUT, SRC])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
