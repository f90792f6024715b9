"""GPU probe (diagnostic build, tools/lab/run_lineq_stamps.sh): where k_fme_batch spends its clock ticks, per phase,
summed over the systems of a batch by lane 0 of each."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from xpoly_amd import _capi
from xpoly_amd.lineq import Lineq
from tools import gen
ctx = xpoly_amd.Context(0)
lq = Lineq(ctx)
lib = _capi.lib()
rng = np.random.default_rng(0)
nb = 4096
names = ["load", "classify", "normalise", "free rows", "pair sums", "reduce", "store", "-", "r:iden", "r:classify",
         "r:tighten", "r:compact", "i:hash", "i:search", "i:compact"]
for rows, nv in ((16, 8), (40, 12), (60, 19)):
    base = np.stack([gen.random_system(rng, rows, nv) for _ in range(256)])
    mats = np.ascontiguousarray(np.tile(base, (nb // 256, 1, 1, 1)))
    out = (C.c_ulonglong * 16)()
    lib.xpg_lineq_debug(ctx._h, out)
    lq.fme(mats, nv, 0)
    lib.xpg_lineq_debug(ctx._h, out)
    tot = sum(out[k] for k in range(7))
    print("fme %dx%d: %.0f ticks per system" % (rows, nv + 1, tot / nb))
    for k, n in enumerate(names):
        if n != "-":
            print("   %-11s %9.0f  %5.1f %%" % (n, out[k] / nb, 100.0 * out[k] / tot))
