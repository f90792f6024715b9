"""GPU probe: per-node latency of a LONE tree walk (one problem per call) for the knapsacks of the MIP leg."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import xpoly_amd
from xpoly_amd.six import mip_batch
from tools import gen
ctx = xpoly_amd.Context(0)
leq, tg = gen.knapsack_batch_rat(1024, 24)
st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq)
# per-problem node counts: solve one by one (also the lone latency)
rows = []
for b in range(0, 1024, 8):
    mip_batch(ctx, True, True, tg[b:b + 1], leq[b:b + 1])
    t0 = time.perf_counter(); r = mip_batch(ctx, True, True, tg[b:b + 1], leq[b:b + 1]); dt = time.perf_counter() - t0
    rows.append((int(r[3]), dt * 1e6))
rows = np.array(rows)
n, t = rows[:, 0], rows[:, 1]
A = np.vstack([n, np.ones_like(n)]).T
slope, icpt = np.linalg.lstsq(A, t, rcond=None)[0]
print("128 problems solved alone: nodes min %d median %d max %d; time = %.1f us + %.1f us per node" % (n.min(), np.median(n), n.max(), icpt, slope))
t0 = time.perf_counter(); st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq); dt = time.perf_counter() - t0
print("whole batch of 1024: %.2f ms, %d nodes" % (dt * 1e3, nodes))
