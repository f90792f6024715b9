"""ONE dependence polyhedron per call (12 x 5) through xpg_dep_is_empty_batch_ex_rat32 and xpg_has_solution_rat32, 300 calls each:
mean latency; under rocprofv3 --kernel-trace --stats the kernels' share of it (tools/lab/run_one_dep.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import xpoly_amd  # noqa: E402
from tools import gen  # noqa: E402
from xpoly_amd.six import dep_is_empty_batch, has_solution  # noqa: E402

ctx = xpoly_amd.Context(0)
rng = np.random.default_rng(5)
sysm = gen.random_system(rng, 12, 4); sysm[..., 1] = 1
vc = gen.to_rat(gen.vc_nonneg(4, False))
one = sysm[None]
for name, fn in (("dep_is_empty(1)", lambda: dep_is_empty_batch(ctx, one)), ("has_solution", lambda: has_solution(ctx, sysm, None, vc, 4, True, True))):
    fn()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    print(name, "%.1f us per call" % ((time.perf_counter() - t0) / 300 * 1e6), flush=True)
