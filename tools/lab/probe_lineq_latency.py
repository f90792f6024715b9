"""GPU probe: latency of ONE-system row-elimination calls through the host-array entry points (what the drop-in
Lineq adapter issues per call)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.lineq import Lineq
from tools import gen
ctx = xpoly_amd.Context(0)
lq = Lineq(ctx)
rng = np.random.default_rng(0)
for rows, nv in ((16, 8), (40, 12)):
    m = gen.random_system(rng, rows, nv)[None]
    for name, fn in (("reduce", lambda: lq.reduce(m, nv, True)), ("fme", lambda: lq.fme(m, nv, 0)), ("rank", lambda: lq.rank(m))):
        fn()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        print("%-6s %2dx%2d: %7.1f us per call" % (name, rows, nv + 1, (time.perf_counter() - t0) / 200 * 1e6))
