"""GPU probe: the MIP leg's batch -- 1024 0-1 knapsacks of 24 variables, the tree of each walked on the device by one
workgroup (k_mip_tree) -- four launches, time and nodes per launch (under rocprofv3 --pmc: tools/lab/run_mip_pmc.sh)."""
import time

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd.six import mip_batch

ctx = xpoly_amd.Context(0)
leq, tg = gen.knapsack_batch_rat(1024, 24)
mip_batch(ctx, True, True, tg, leq)
for rep in range(4):
    t0 = time.perf_counter()
    st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq)
    dt = time.perf_counter() - t0
    print("launch %d: ms %.3f nodes %d solved %d" % (rep, dt * 1e3, nodes, int((st == 0).sum())))
