"""Builds xpoly_amd/libxpoly_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

-ffp-contract=off is part of the numerical contract, not a tuning flag: the
reference rounds after the multiply and again after the add (lpsol.h:1485-1489).

csrc/xpoly_amd.hip is compiled four times in parallel (-DXPG_PART=0..3, each part
holding the entry points of one subsystem and including only the kernel headers
they launch) and the objects are linked into one shared object: the device code
of all kernels in one translation unit took 4.7 minutes, the parts take about
the time of the largest. XPG_BUILD_MONOLITHIC=1 builds the single-TU form.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "xpoly_amd.hip")
OUT = os.path.join(HERE, "libxpoly_amd.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
# the same sources with -DXPG_TEST_HOOKS: fault injection, forced routes, debug prints and the lab's A/B knobs are compiled
# in (scalar.hip.h xpg_hook). Loaded only by the tests that need a hook (tests/conftest.py needs_hooks) and by tools/lab.
HOOKS_OUT = os.path.join(HERE, "libxpoly_amd_hooks.so")
HOOKS_OBJ = os.path.join(HERE, "csrc", "_obj", "hooks")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17"]
PARTS = 4
# headers each part includes (directly or not): a part is recompiled when one of them, or xpoly_amd.hip, is newer than
# its object
COMMON = ["xpoly_amd.hip", "scalar.hip.h", "ctx.hip.h", "rat_ops.hip.h", "../../include/xpoly_amd.h"]
DEPS = {
    0: ["lp_kernels.hip.h", "lp_pipe_r32.hip.h", "lp_fused_r32.hip.h", "lp_host.hip.h", "lp_blocked.hip.h", "lp_chain.hip.h", "warm_mip.hip.h", "warm_mip_batch.hip.h"],
    1: ["lp_kernels.hip.h", "six_host.hip.h", "batch_kernels.hip.h"],
    2: ["lp_kernels.hip.h", "six_host.hip.h", "batch_kernels.hip.h", "lineq_shared.hip.h", "mip_host.hip.h", "mip_kernels.hip.h"],
    3: ["lineq_shared.hip.h", "lineq_host.hip.h", "lineq_kernels.hip.h"],
}


def sources():
    d = os.path.join(HERE, "csrc")
    inc = os.path.join(HERE, "..", "include", "xpoly_amd.h")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f))] + [inc]


def stale(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in sources())


def build_hooks(force=False):
    """xpoly_amd/libxpoly_amd_hooks.so (-DXPG_TEST_HOOKS)."""
    if not force and not stale(HOOKS_OUT):
        return HOOKS_OUT
    return build(force=force, extra=["-DXPG_TEST_HOOKS"], out=HOOKS_OUT, obj=HOOKS_OBJ)


def build_all(force=False):
    """The product library and the test-hooks variant, compiled side by side."""
    with ThreadPoolExecutor(max_workers=2) as ex:
        a = ex.submit(build, force)
        b = ex.submit(build_hooks, force)
        return a.result(), b.result()


def build(force=False, extra=(), out=None, obj=None):
    """force: recompile every part from its source whatever the time stamps say (what __graft_entry__.build() asks for:
    the driver's build step proves the source, it does not re-link yesterday's objects).
    out / obj: another library path and object directory (diagnostic builds: -DXPG_STAMPS, -DXPG_LIFE)."""
    if out is None and not force and not stale():
        return OUT
    OUT_ = out or OUT
    OBJ_ = obj or OBJ
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    extra = list(extra) + os.environ.get("XPG_BUILD_FLAGS", "").split()
    if os.environ.get("XPG_BUILD_MONOLITHIC") == "1":
        subprocess.check_call([hipcc] + CFLAGS + extra + ["-shared", "-o", OUT_, SRC])
        return OUT_
    os.makedirs(OBJ_, exist_ok=True)
    objs = [os.path.join(OBJ_, "part%d.o" % p) for p in range(PARTS)]

    flags_key = " ".join(CFLAGS + extra)
    key_file = os.path.join(OBJ_, "flags.txt")
    same_flags = os.path.exists(key_file) and open(key_file).read() == flags_key

    def part_stale(p):
        if force or not same_flags or not os.path.exists(objs[p]):
            return True
        t = os.path.getmtime(objs[p])
        d = os.path.join(HERE, "csrc")
        return any(os.path.getmtime(os.path.join(d, f)) > t for f in COMMON + DEPS[p])

    def one(p):
        if part_stale(p):
            subprocess.check_call([hipcc] + CFLAGS + extra + ["-DXPG_PART=%d" % p, "-c", SRC, "-o", objs[p]])

    with ThreadPoolExecutor(max_workers=PARTS) as ex:
        list(ex.map(one, range(PARTS)))
    with open(key_file, "w") as f:
        f.write(flags_key)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", OUT_] + objs)
    return OUT_


if __name__ == "__main__":
    import sys
    # `python -m xpoly_amd.build` recompiles the parts whose sources changed; `--force` all four
    if "--hooks" in sys.argv[1:]:
        print(build_hooks(force="--force" in sys.argv[1:]))
    elif "--product" in sys.argv[1:]:
        print(build(force="--force" in sys.argv[1:]))
    else:
        print(*build_all(force="--force" in sys.argv[1:]))
