"""GPU diagnostic: whole solves (SIX::TwoStageMethod on a device-resident LP) of many small and mid-size fp64 LPs through the
blocked loop (chain launch + pass) and the pipelined loop (two launches per pivot) -- where should XPG_LOOP's automatic choice
switch (lp_host.hip.h: Lp::queue_iterations)? Prints per family the total time of each loop and the worst single ratio."""
import os
import time

import numpy as np

import xpoly_amd
from tools import gen

F64 = 0
rng = np.random.default_rng(23)
cases = []
for (m, n) in ((24, 40), (48, 64), (96, 80), (150, 120), (300, 200), (200, 600), (600, 500)):
    for kind in range(3):
        for rep in range(3):
            p = gen.random_problem(rng, F64, kind, m, n, plain=True)
            cases.append(("%dx%d kind %d" % (m, n, kind), p["leq"], p["tgtf"]))
    cases.append(("%dx%d hard" % (m, n), *gen.hard_lp_f64(m, n)))
    cases.append(("%dx%d dense" % (m, n), *gen.dense_lp_f64(m, n)))
res = {}
for mode in ("block", "pipe"):
    os.environ["XPG_LOOP"] = mode
    ctx = xpoly_amd.Context(0)
    for k, (name, leq, tg) in enumerate(cases):
        best = None
        for rep in range(2):
            lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
            t0 = time.perf_counter()
            st = lp.two_stage()
            dt = time.perf_counter() - t0
            piv = lp.pivots_done()
            lp.close()
            best = dt if best is None else min(best, dt)
        res.setdefault(k, {})[mode] = (best, piv, st)
    ctx.close()
fam = {}
for k, (name, _, _) in enumerate(cases):
    b, p = res[k]["block"], res[k]["pipe"]
    assert b[1:] == p[1:], (name, b, p)
    f = fam.setdefault(name.split()[0], [0.0, 0.0, 0, 0.0, ""])
    f[0] += b[0]; f[1] += p[0]; f[2] += b[1]
    if b[0] / p[0] > f[3]: f[3] = b[0] / p[0]; f[4] = "%s: %d pivots, block %.2f ms, pipe %.2f ms" % (name, b[1], b[0] * 1e3, p[0] * 1e3)
for shape, (tb, tp, piv, worst, wname) in fam.items():
    print("%-9s %7d pivots  block %9.2f ms  pipe %9.2f ms  (block / pipe %.2f)   worst case for block: %.2f (%s)" % (shape, piv, tb * 1e3, tp * 1e3, tb / tp, worst, wname))
