"""One mid-size whole solve (hard 300 x 300: the oracle says status 1 after 214 796 pivots) through the
blocked loop under the A/B knobs given as KEY=VALUE arguments groups separated by '/'."""
import os
import sys
import time

import xpoly_amd
from tools import gen

F64 = 0
leq, tg = gen.hard_lp_f64(300, 300)
groups = " ".join(sys.argv[1:]).split("/") if len(sys.argv) > 1 else [""]
for g in groups:
    env = dict(kv.split("=") for kv in g.split())
    for k in ("XPG_LOOP", "XPG_BLOCK", "XPG_BLK_ROWS", "XPG_BLK_TPB_PICK", "XPG_BLK_TPB_PREP"):
        os.environ.pop(k, None)
    os.environ["XPG_LOOP"] = "block"
    os.environ.update(env)
    ctx = xpoly_amd.Context(0)
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    t0 = time.perf_counter()
    st = lp.two_stage()
    dt = time.perf_counter() - t0
    print("%-60s status %d, %d pivots, %.0f ms" % (g or "(default)", st, lp.pivots_done(), dt * 1e3), flush=True)
    lp.close(); ctx.close()
