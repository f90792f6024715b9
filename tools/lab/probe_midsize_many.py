"""Whole solves of assorted mid-size fp64 LPs: blocked loop against pipelined loop (status, pivots,
CRCs of tableau / objective row / basis / trace). A guard against state races that only long runs
with many closed batches expose (see tests/test_gpu_edges.py::test_midsize_whole_solve_every_loop)."""
import os
import sys
import zlib

import numpy as np

import xpoly_amd
from tools import gen

F64 = 0
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(seed)
cases = []
for _ in range(4):
    m, n = int(rng.integers(80, 420)), int(rng.integers(80, 520))
    cases.append(("hard %dx%d" % (m, n), gen.hard_lp_f64(m, n)))
for _ in range(3):
    m, n = int(rng.integers(60, 300)), int(rng.integers(60, 400))
    cases.append(("dense %dx%d" % (m, n), gen.dense_lp_f64(m, n)))
for _ in range(5):
    p = gen.random_problem(rng, F64, int(rng.integers(0, 3)), int(rng.integers(60, 260)), int(rng.integers(60, 260)), plain=True)
    cases.append(("random %dx%d" % p["leq"].shape, (p["leq"], p["tgtf"])))
bad = 0
for name, (leq, tg) in cases:
    keys = {}
    for mode in ("pipe", "block"):
        os.environ["XPG_LOOP"] = mode
        ctx = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
        st = lp.two_stage()
        key = [st, lp.pivots_done()]
        if st != 2:
            out = lp.read()
            key += [zlib.crc32(out["tab"].tobytes()), zlib.crc32(out["tgtf"].tobytes()), zlib.crc32(np.asarray(out["eq2bv"]).tobytes())]
        key.append(zlib.crc32(lp.trace().tobytes()))
        keys[mode] = key
        lp.close(); ctx.close()
    ok = keys["pipe"] == keys["block"]
    bad += not ok
    print("%-18s %s %s" % (name, "OK      " if ok else "MISMATCH", keys["block"][:2] if ok else keys), flush=True)
print("midsize whole solves: %d cases, %d mismatches" % (len(cases), bad))
