"""GPU probe: wall time of consecutive xpg_lp_iterate chunks (with a sync after each) to locate
intermittent stalls of the queued loop. usage: python -m tools.probe_stall [chunk] [count]"""
import sys
import time

import xpoly_amd
from tools import gen

F64 = 0
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 400
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
for rep in range(3):
    lp.begin()
    ctx.sync()
    ts = []
    for c in range(count):
        t0 = time.perf_counter()
        lp.iterate(chunk)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("rep %d: ms per %d-pivot chunk:" % (rep, chunk), " ".join("%.1f" % t for t in ts))
