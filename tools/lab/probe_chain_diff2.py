"""GPU diagnostic: batch by batch, where does the tableau of the chain-kernel loop first differ from the
launch-per-stage loop's?"""
import os
import sys

import numpy as np

import xpoly_amd
from tools import gen

F64 = 0
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 300)
leq, tg = gen.hard_lp_f64(m, n)
os.environ["XPG_LOOP"] = "block"
lps = {}
for name, ch in (("chain", "1"), ("launch", "0")):
    os.environ["XPG_CHAIN"] = ch
    c = xpoly_amd.Context(0)
    lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
    lp.begin()
    lps[name] = (c, lp)
total = 0
while total < 2048:
    for k, (c, lp) in lps.items():
        lp.iterate(16)
    total += 16
    ra = lps["chain"][1].read(); rb = lps["launch"][1].read()
    ta, tb = lps["chain"][1].trace(), lps["launch"][1].trace()
    dt = ra["tab"].view(np.uint64) != rb["tab"].view(np.uint64)
    do = ra["tgtf"].view(np.uint64) != rb["tgtf"].view(np.uint64)
    if dt.any() or do.any() or not np.array_equal(ta, tb):
        print("after %d iterations (pivots %d / %d): %d tableau cells differ, %d objective entries" % (total, len(ta), len(tb), dt.sum(), do.sum()))
        rows, cols = np.nonzero(dt)
        print("columns with differences:", np.unique(cols)[:20].tolist(), "rows:", len(np.unique(rows)))
        print("objective entries differing:", np.nonzero(do)[0][:20].tolist())
        print("last pivots chain :", ta[-16:].tolist())
        print("last pivots launch:", tb[-16:].tolist())
        for key in ("nvset", "bvset", "bv2eq", "eq2bv"):
            if not np.array_equal(ra[key], rb[key]):
                print(key, "differs at", np.nonzero(ra[key] != rb[key])[0][:10].tolist())
        break
else:
    print("no difference in %d iterations" % total)
