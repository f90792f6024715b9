"""GPU probe: the warm-started branch and bound (N4, non-parity) on multi-constraint knapsacks: nodes, dual pivots per
node against the root's primal pivots, wall time; HiGHS for the optimum."""
import time
import numpy as np
import xpoly_amd
from scipy.optimize import Bounds, LinearConstraint, milp
from xpoly_amd.six import mip_warm

ctx = xpoly_amd.Context(0)
for nv, m, seed in ((16, 4, 1), (24, 6, 2), (32, 8, 3), (40, 10, 4)):
    rng = np.random.default_rng(seed)
    A = rng.integers(1, 20, size=(m, nv)).astype(np.float64)
    b = np.floor(A.sum(axis=1) * 0.4)
    c = rng.integers(5, 30, size=nv).astype(np.float64)
    Ab = np.concatenate([A, np.eye(nv)], axis=0); bb = np.concatenate([b, np.ones(nv)])
    leq = np.concatenate([Ab, bb[:, None]], axis=1); tgtf = np.concatenate([c, [0.0]])
    ref = milp(c=-c, constraints=LinearConstraint(Ab, -np.inf, bb), integrality=np.ones(nv), bounds=Bounds(0, np.inf))
    mip_warm(ctx, True, tgtf, leq, True)
    t0 = time.perf_counter(); st, v, sol, stats = mip_warm(ctx, True, tgtf, leq, True); dt = time.perf_counter() - t0
    print("0-1 knapsack %2d vars x %2d rows: status %d, optimum %.0f (HiGHS %.0f), %d nodes, depth %d, root %d primal pivots, "
          "%.2f dual pivots per node, %.1f ms (%.0f us per node)" % (nv, m, st, v, -ref.fun, stats["nodes"], stats["max_depth"],
          stats["root_pivots"], stats["dual_pivots"] / max(1, stats["nodes"] - 1), dt * 1e3, dt * 1e6 / stats["nodes"]))
