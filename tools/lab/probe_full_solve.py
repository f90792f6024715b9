"""GPU one-off: the bench LP (4096 x 4095) solved to the reference's end state (4165 pivots) by the
blocked and the pipelined loop: status, trace and every bit of the final tableau must agree; twice each
(determinism)."""
import os
import zlib

import numpy as np

import xpoly_amd
from tools import gen

F64 = 0
leq, tg = gen.hard_lp_f64(4096, 4095)
seen = {}
for mode in ("block", "pipe", "block", "pipe"):
    os.environ["XPG_LOOP"] = mode
    ctx = xpoly_amd.Context(0)
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    st = lp.two_stage()
    out = lp.read()
    tr = lp.trace()
    key = (st, lp.pivots_done(), zlib.crc32(out["tab"].tobytes()), zlib.crc32(out["tgtf"].tobytes()), zlib.crc32(tr.tobytes()))
    print(mode, key)
    seen.setdefault("all", key)
    assert key == seen["all"], "loops disagree"
    lp.close(); ctx.close()
print("full solve: blocked == pipelined, deterministic")
