"""GPU diagnostic: first pivot at which the blocked loop with the chain kernel (stages 1.. of a batch in one
launch) departs from the launch-per-stage blocked loop, on the hard 300 x 300 LP."""
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
step, total = 64, 0
while total < 4096:
    sts = {k: lp.iterate(step) for k, (c, lp) in lps.items()}
    total += step
    tr = {k: lp.trace() for k, (c, lp) in lps.items()}
    a, b = tr["chain"], tr["launch"]
    k = min(len(a), len(b))
    same = np.array_equal(a[:k], b[:k]) and len(a) == len(b)
    if not same or sts["chain"] != sts["launch"]:
        d = next((q for q in range(k) if not np.array_equal(a[q], b[q])), k)
        print("after %d iterations: status %s, pivots %d / %d, first differing pivot %d" % (total, sts, len(a), len(b), d))
        print("chain :", a[max(0, d - 3): d + 4].tolist())
        print("launch:", b[max(0, d - 3): d + 4].tolist())
        ra = lps["chain"][1].read(); rb = lps["launch"][1].read()
        for key in ("nvset", "bvset", "bv2eq", "eq2bv"):
            print(key, "equal" if np.array_equal(ra[key], rb[key]) else "DIFFERENT")
        break
else:
    print("no difference in %d iterations" % total)
