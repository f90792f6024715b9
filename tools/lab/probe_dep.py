"""GPU probe: the dependence-test front end (DepPoly::is_empty) and batched 0-1 MIPs (config 5)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.six import dep_is_empty_batch, mip_batch
from tools import gen
ctx = xpoly_amd.Context(0)
rng = np.random.default_rng(0)
for (rows, nv, nb) in ((12, 4, 4096), (24, 8, 4096)):
    base = np.stack([gen.random_system(rng, rows, nv) for _ in range(512)]); base[..., 1] = 1
    mats = np.ascontiguousarray(np.tile(base, (nb // 512, 1, 1, 1)))
    dep_is_empty_batch(ctx, mats[:64])
    t0 = time.perf_counter(); empty, nodes = dep_is_empty_batch(ctx, mats); dt = time.perf_counter() - t0
    print("is_empty %dx%d: %d polyhedra in %.1f ms -> %.0f polyhedra/s, %d node LPs (%.0f nodes/s), %d empty, %d undefined"
          % (rows, nv + 1, nb, dt * 1e3, nb / dt, nodes, nodes / dt, int((empty == 1).sum()), int((empty < 0).sum())))
for nv in (8, 16, 31):
    nb = 1024
    probs = []
    for _ in range(128):
        p = gen.random_mip(rng, 1, nv, False)
        ub = np.zeros((nv, nv + 1), dtype=np.int32); ub[np.arange(nv), np.arange(nv)] = 1; ub[:, nv] = 1
        # one knapsack row + x <= 1 rows: keep rows <= cols so that the reference is defined
        p["leq"] = np.concatenate([p["leq"], gen.to_rat(ub)], axis=0); probs.append(p)
    leq = np.ascontiguousarray(np.tile(np.stack([p["leq"] for p in probs]), (nb // 128, 1, 1, 1)))
    tg = np.ascontiguousarray(np.tile(np.stack([p["tgtf"] for p in probs]), (nb // 128, 1, 1)))
    mip_batch(ctx, True, True, tg[:32], leq[:32])
    t0 = time.perf_counter(); st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq); dt = time.perf_counter() - t0
    print("0-1 MIP %d vars: %d problems in %.1f ms -> %.0f MIPs/s, %d nodes (%.0f nodes/s), status hist %s"
          % (nv, nb, dt * 1e3, nb / dt, nodes, nodes / dt, np.bincount(np.clip(st, 0, 4), minlength=4).tolist()))
