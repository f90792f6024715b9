"""GPU probe: config 4 -- rational tableau 1024 x 2048, K = 16 pivots (crosses the first appro)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
m, n = 1024, 1023           # slack tableau 1024 x 2048
leq, tg = gen.int_lp_rat(m, n)
for rep in range(2):
    lp = xpoly_amd.DeviceLP(ctx, 1, leq, tg)
    lp.begin()
    ctx.sync()
    ctx.profile_begin(16, 1)
    t0 = time.perf_counter()
    st = lp.iterate(16)
    ctx.sync()
    dt = time.perf_counter() - t0
    n_ev, ms = ctx.profile_end()
    print("rational 1024x2048: 16 pivots in %.2f ms -> %.1f pivots/s; sweep avg %.1f us (%d launches) -> %.1f GB/s algorithmic (bound is integer ALU, not HBM); status %d"
          % (dt * 1e3, 16 / dt, ms / max(n_ev, 1) * 1e3, n_ev, 2 * 1024 * 2048 * 8 / (ms / max(n_ev, 1) / 1e3) / 1e9, st))
    lp.close()
