"""GPU probe: the batched warm-started branch and bound (xpg_mip_warm_batch_f64) on the MIP leg's knapsacks: time, nodes/s,
dual pivots per node, and the optima against the one-tree form on a sample."""
import time

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd.six import mip_warm_batch

ctx = xpoly_amd.Context(0)
for nb in (1024, 8192):
    leq_r, tg_r = gen.knapsack_batch_rat(nb, 24)
    leq = leq_r[..., 0].astype(np.float64); tg = tg_r[..., 0].astype(np.float64)
    mip_warm_batch(ctx, True, tg, leq, is_bin=True)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        st, v, sol, stats = mip_warm_batch(ctx, True, tg, leq, is_bin=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("nb %5d: %.2f ms, %.0f MIPs/s, %.2f M nodes/s, %.2f dual pivots per node, solved %d, checksum %.6f" % (
        nb, best * 1e3, nb / best, stats["nodes"] / best / 1e6, stats["dual_pivots"] / max(1, stats["nodes"] - nb), int((st == 0).sum()), float(v.sum())))
