"""GPU diagnostic (-DXPG_STAMPS build, tools/lab/run_mip_stamps.sh): where a node of the device-side tree walk spends
its time -- rebuilding the node problem, the LDS solve, the recursion's feed-back."""
import ctypes as C
import xpoly_amd
from xpoly_amd._capi import lib
from xpoly_amd.six import mip_batch
from tools import gen
ctx = xpoly_amd.Context(0)
leq, tg = gen.knapsack_batch_rat(1024, 24)
d = (C.c_ulonglong * 4)()
e = (C.c_ulonglong * 8)()
mip_batch(ctx, True, True, tg, leq)
lib().xpg_mip_debug(ctx._h, d); lib().xpg_lp_solve_debug(ctx._h, e)
st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq)
lib().xpg_mip_debug(ctx._h, d); lib().xpg_lp_solve_debug(ctx._h, e)
for k, n in enumerate(("build node", "LDS solve", "feed-back")):
    print("%-10s %7.1f us per node" % (n, d[k] * 0.01 / nodes))
print("inside the solve: %d node LPs through phase one (%.1f us each), %d straight to the slack form (%.1f us build each);"
      % (e[4], e[0] * 0.01 / max(1, e[4]), e[5], e[1] * 0.01 / max(1, e[5])))
print("main loops %.1f us per node, %.1f pivots per node, %.2f us per pivot" % (e[2] * 0.01 / nodes, e[3] / nodes, e[2] * 0.01 / max(1, e[3])))
