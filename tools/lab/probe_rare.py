"""Probe: how often does the pipelined device loop take a non-pivot iteration (deferred rare
branch) on dependence-test-like LPs? pivots_done < iterations means such iterations happened."""
import numpy as np
import xpoly_amd
from tools import gen

F64 = 0
ctx = xpoly_amd.Context(0)
leqs, tgs = gen.small_lp_batch_f64(8, 24, 33, family=1, seed=gen.XS_SEED + 77)
leqs[:, :, -1] = np.abs(leqs[:, :, -1])          # origin feasible: the slack branch of stage 1
for b in range(8):
    lp = xpoly_amd.DeviceLP(ctx, F64, leqs[b], tgs[b])
    lp.begin()
    its = 0
    st = xpoly_amd.six.XPG_RUNNING
    while st == xpoly_amd.six.XPG_RUNNING and its < 20000:
        st = lp.iterate(100); its += 100
    print("lp", b, "status", st, "iterations<=", its, "pivots", lp.pivots_done())
    lp.close()
