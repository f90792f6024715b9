"""GPU diagnostic (-DXPG_STAMPS build, tools/run_mip_stamps.sh): where a node of the device-side tree walk spends
its time -- rebuilding the node problem, the LDS solve, the recursion's feed-back."""
import ctypes as C
import xpoly_amd
from xpoly_amd._capi import lib
from xpoly_amd.six import mip_batch
from tools import gen
ctx = xpoly_amd.Context(0)
leq, tg = gen.knapsack_batch_rat(1024, 24)
d = (C.c_ulonglong * 4)()
mip_batch(ctx, True, True, tg, leq)
lib().xpg_mip_debug(ctx._h, d)
st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq)
lib().xpg_mip_debug(ctx._h, d)
for k, n in enumerate(("build node", "LDS solve", "feed-back")):
    print("%-10s %7.1f us per node" % (n, d[k] * 0.01 / nodes))
