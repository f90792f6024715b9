"""GPU diagnostic: pivots/s of the blocked loop (chain launch + sweep) at smaller shapes than the bench's, for A/B runs
of two builds (XPG_SO_PATH) -- does the chain workers' first sleep (lp_chain.hip.h, CH_NAP_*) cost anything where a
stage is shorter?"""
import os
import time

import xpoly_amd
from tools import gen

os.environ["XPG_LOOP"] = "block"
ctx = xpoly_amd.Context(0)
out = []
for m, n, k in ((512, 1023, 1200), (1024, 1535, 1200), (2048, 4095, 1200), (4096, 4095, 1200)):
    leq, tg = gen.hard_lp_f64(m, n)
    best = 0.0
    for rep in range(3):
        lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
        lp.begin(); lp.iterate(240)
        p0 = lp.pivots_done()
        t0 = time.perf_counter()
        lp.iterate(k)
        dt = time.perf_counter() - t0
        best = max(best, (lp.pivots_done() - p0) / dt)
        lp.close()
    out.append("%dx%d %.1f k" % (m, n + 1, best / 1e3))
print(os.path.basename(os.environ.get("XPG_SO_PATH", "tree")), " | ".join(out))
