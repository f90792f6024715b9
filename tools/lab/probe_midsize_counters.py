"""GPU diagnostic: what the blocked loop's batches look like on mid-size LPs -- pivots per sweep, full / partial sweeps, chain
launches (with stage 0 inside: folds), time per pivot."""
import os
import time

import xpoly_amd
from tools import gen

os.environ["XPG_LOOP"] = "block"
ctx = xpoly_amd.Context(0)
for name, (leq, tg) in (("hard 300x300", gen.hard_lp_f64(300, 300)), ("dense 300x400", gen.dense_lp_f64(300, 400)), ("hard 120x200", gen.hard_lp_f64(120, 200)),
                        ("hard 600x500", gen.hard_lp_f64(600, 500)), ("hard 1024x1023", gen.hard_lp_f64(1024, 1023))):
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
    t0 = time.perf_counter()
    st = lp.two_stage(60000)
    dt = time.perf_counter() - t0
    piv = lp.pivots_done()
    full, part = lp.counters()
    lp.chain_aborts()
    print("%-15s status %d, %6d pivots in %7.1f ms = %.2f us/pivot; sweeps %d full + %d partial = %.1f pivots per sweep; chain launches %d, with stage 0 inside %d"
          % (name, st, piv, dt * 1e3, dt * 1e6 / piv, full, part, piv / max(1, full + part), lp.chain_runs, lp.chain_folds()))
    lp.close()
