import os, zlib, numpy as np, xpoly_amd
from tools import gen
F64 = 0
for (m, n) in ((6000, 5999), (1500, 9000), (5000, 700)):
    leq, tg = gen.hard_lp_f64(m, n)
    keys = []
    for mode in ("block", "pipe"):
        os.environ["XPG_LOOP"] = mode
        ctx = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
        lp.begin()
        st = lp.iterate(237)
        out = lp.read()
        keys.append((st, lp.pivots_done(), zlib.crc32(out["tab"].tobytes()), zlib.crc32(out["tgtf"].tobytes()), zlib.crc32(lp.trace().tobytes())))
        lp.close(); ctx.close()
    print(m, n, keys[0], "OK" if keys[0] == keys[1] else "MISMATCH %s" % (keys[1],))
