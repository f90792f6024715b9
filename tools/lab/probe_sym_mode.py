"""Throughput of the opt-in symbols-as-variables mode of the dependence test (host controller: the widened system has free
variables) beside the default device path on systems without symbols."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import xpoly_amd  # noqa: E402
from tools import gen  # noqa: E402
from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_batch_symbols_as_vars, six_last_profile  # noqa: E402

ctx = xpoly_amd.Context(0)
rng = np.random.default_rng(1)
for nb in (64, 512):
    mats = np.stack([gen.random_system(rng, 9, 5) for _ in range(nb)]); mats[..., 1] = 1
    dep_is_empty_batch_symbols_as_vars(ctx, mats[:8], 3)
    t0 = time.perf_counter(); e, nodes = dep_is_empty_batch_symbols_as_vars(ctx, mats, 3); dt = time.perf_counter() - t0
    print("symbols as variables: %d polyhedra (3 variables + 2 symbols, 9 rows) %.1f ms = %.0f polyhedra/s, nodes %d, answers %s"
          % (nb, dt * 1e3, nb / dt, nodes, np.bincount(e + 7, minlength=9).tolist()), flush=True)
    dep_is_empty_batch(ctx, mats[:8])
    t0 = time.perf_counter(); e, nodes = dep_is_empty_batch(ctx, mats); dt = time.perf_counter() - t0
    print("the same systems read as 5 variables, no symbols (device path): %.1f ms = %.0f polyhedra/s" % (dt * 1e3, nb / dt), flush=True)
