"""GPU probe: the bench LP (4096 x 4095, tableau 4096 x 8192 fp64) to its optimum in the parity
mode (the reference's first-positive pricing) and in the opt-in non-parity mode (Dantzig pricing,
tolerant is_feasible): status, pivots, wall time, objective."""
import time

import xpoly_amd
from tools import gen

F64 = 0
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
for name, opts in (("parity", None), ("dantzig+tolerant", (1, 1e-9))):
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    if opts:
        lp.set_options(*opts)
    ctx.sync()
    t0 = time.perf_counter()
    st = lp.two_stage()
    ctx.sync()
    dt = time.perf_counter() - t0
    out = lp.read(want_tab=False)
    print("%-18s status %d  pivots %5d  %.3f s  (%.0f pivots/s)  objective %.12g"
          % (name, st, lp.pivots_done(), dt, lp.pivots_done() / dt, out["maxv"]))
    lp.close()
