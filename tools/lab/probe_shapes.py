"""One LP shape per process through the default device loop: python tools/lab/probe_shapes.py M N [K] -- prints loop info, status and
pivots/s (a memory fault takes the process down: run_shapes.sh runs the shapes one after the other)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import xpoly_amd  # noqa: E402
from tools import gen  # noqa: E402

m, n = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(m, n)
lp = xpoly_amd.DeviceLP(ctx, 0, leq, tg)
lp.begin()
print(m, n, lp.loop_info(), flush=True)
t0 = time.perf_counter()
st = lp.iterate(K)
dt = time.perf_counter() - t0
print(m, n, "status", st, "pivots", lp.pivots_done(), "%.1f pivots/s" % (K / dt), "aborts", lp.chain_aborts(), flush=True)
