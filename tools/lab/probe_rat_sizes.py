"""GPU probe: exact pivots/s of the device-resident Rational loop at several tableau sizes (XPG_R32_LOOP=pipe for the
two-launch loop)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
SIZES = ((64, 63, 48), (256, 255, 32), (512, 700, 24), (1024, 1023, 16), (2048, 1500, 12))
if len(sys.argv) > 1 and sys.argv[1] == 'more':
    SIZES = ((128, 200, 40), (384, 400, 32), (384, 900, 24), (768, 1000, 16), (1280, 1000, 16), (1536, 1100, 12), (1536, 2500, 12))
for m, n, K in SIZES:
    leq, tg = gen.int_lp_rat(m, n)
    lp = xpoly_amd.DeviceLP(ctx, 1, leq, tg)
    best = None
    for rep in range(4):
        lp.begin(); ctx.sync()
        t0 = time.perf_counter(); lp.iterate(K); dt = time.perf_counter() - t0
        if rep: best = dt if best is None else min(best, dt)
    print("%5d x %5d  K %3d: %7.2f us per pivot" % (m, n + m + 1, K, best / K * 1e6))
    lp.close()
