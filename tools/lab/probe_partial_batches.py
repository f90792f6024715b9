"""GPU probe: how often does the blocked loop sweep a batch of fewer than 16 pivots when it runs
whole solves -- i.e. how often does a rare branch of SIX::solveSlackForm close a batch early?"""
import os
import time

import xpoly_amd
from tools import gen

os.environ["XPG_LOOP"] = "block"
ctx = xpoly_amd.Context(0)
for m, n in ((4096, 4095), (1024, 1500), (300, 300)):
    leq, tg = gen.hard_lp_f64(m, n)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
    t0 = time.perf_counter()
    st = lp.two_stage()
    dt = time.perf_counter() - t0
    f, p = lp.counters()
    piv = lp.pivots_done()
    print("hard LP %4d x %4d: status %d after %7d pivots; sweeps: %6d full, %5d partial (%.1f %% of sweeps, %.1f %% of pivots in partial batches)"
          % (m, n, st, piv, f, p, 100.0 * p / max(1, f + p), 100.0 * (piv - 16 * f) / max(1, piv)), "| %.0f pivots/s" % (piv / dt))
    lp.close()
