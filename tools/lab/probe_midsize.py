"""Whole solves (SIX::TwoStageMethod) of mid-size fp64 LPs through the blocked and the pipelined loop:
wall time, pivots, status. Decides where XPG_LOOP's automatic choice should switch (lp_host.hip.h)."""
import os
import time

import numpy as np

import xpoly_amd
from tools import gen

F64 = 0
rng = np.random.default_rng(11)
cases = [("hard 300x300", gen.hard_lp_f64(300, 300)), ("hard 120x200", gen.hard_lp_f64(120, 200)),
         ("dense 300x400", gen.dense_lp_f64(300, 400)), ("dense 64x96", gen.dense_lp_f64(64, 96))]
for k in range(3):
    p = gen.random_problem(rng, F64, int(rng.integers(0, 3)), 150 + 50 * k, 120 + 40 * k, plain=True)
    cases.append(("random %dx%d" % p["leq"].shape, (p["leq"], p["tgtf"])))
for name, (leq, tg) in cases:
    line = []
    for mode in ("block", "pipe", "auto"):
        if mode == "auto": os.environ.pop("XPG_LOOP", None)
        else: os.environ["XPG_LOOP"] = mode
        ctx = xpoly_amd.Context(0)
        best = None
        for rep in range(3):
            lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
            t0 = time.perf_counter()
            st = lp.two_stage()
            dt = time.perf_counter() - t0
            piv = lp.pivots_done()
            lp.close()
            best = dt if best is None else min(best, dt)
        ctx.close()
        line.append("%s: status %d, %d pivots, %.2f ms (%.1f us/pivot)" % (mode, st, piv, best * 1e3, best * 1e6 / max(1, piv)))
    print("%-16s %s | %s | %s" % (name, line[0], line[1], line[2].split(",")[-1]))
