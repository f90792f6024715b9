"""GPU probe: throughput of the wave-per-system row-elimination kernels at dependence-test sizes."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.lineq import Lineq
from tools import gen
ctx = xpoly_amd.Context(0)
lq = Lineq(ctx)
rng = np.random.default_rng(0)
nb = 16384
for rows, nv in ((16, 8), (40, 12), (60, 19)):
    base = np.stack([gen.random_system(rng, rows, nv) for _ in range(256)])
    mats = np.ascontiguousarray(np.tile(base, (nb // 256, 1, 1, 1)))
    for name, fn in (("reduce", lambda: lq.reduce(mats, nv, True)), ("fme", lambda: lq.fme(mats, nv, 0)),
                     ("rank", lambda: lq.rank(mats))):
        fn()
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        print("%-6s %2dx%2d: %8.0f systems/s (incl. PCIe staging of %d systems)" % (name, rows, nv + 1, nb / dt, nb))
