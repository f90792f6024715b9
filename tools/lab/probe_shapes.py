"""Cross-checks the device-resident fp64 loops against each other on large tableaux of several shapes
(status, pivot count, CRCs of tableau / objective row / trace after `PROBE_PIVOTS` iterations).
PROBE_MODES=block,pipe picks the loops (XPG_LOOP values); the first one is the reference."""
import os
import time
import zlib

import xpoly_amd
from tools import gen

F64 = 0
modes = os.environ.get("PROBE_MODES", "block,pipe").split(",")
pivots = int(os.environ.get("PROBE_PIVOTS", "237"))
shapes = [tuple(int(x) for x in s.split("x")) for s in
          os.environ.get("PROBE_SHAPES", "6000x5999,1500x9000,5000x700").split(",")]
for (m, n) in shapes:
    leq, tg = gen.hard_lp_f64(m, n)
    keys = []
    for mode in modes:
        os.environ["XPG_LOOP"] = mode
        ctx = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
        lp.begin()
        t0 = time.perf_counter()
        st = lp.iterate(pivots)
        dt = time.perf_counter() - t0
        out = lp.read()
        keys.append((st, lp.pivots_done(), zlib.crc32(out["tab"].tobytes()), zlib.crc32(out["tgtf"].tobytes()),
                     zlib.crc32(lp.trace().tobytes())))
        print("  %-6s %s  %.1f us/pivot" % (mode, keys[-1], dt * 1e6 / max(1, lp.pivots_done())))
        lp.close(); ctx.close()
    print(m, n, "OK" if all(k == keys[0] for k in keys) else "MISMATCH")
