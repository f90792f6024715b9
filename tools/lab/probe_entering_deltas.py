"""GPU probe: first(t+1) - first(t), the step of the entering column from one pivot to the next, over the bench LPs' first 3840
pivots -- which columns a pick worker could replay ahead of time while it waits for the stage's partials."""
import collections

import numpy as np

import xpoly_amd
from tools import gen

ctx = xpoly_amd.Context(0)
for (m, n) in ((4096, 4095), (4096, 8192)):
    leq, tg = gen.hard_lp_f64(m, n)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
    del leq
    lp.begin(); lp.iterate(3840)
    e = lp.trace()[:, 0].astype(np.int64)
    d = np.diff(e)
    c = collections.Counter(d.tolist())
    tot = len(d)
    top = c.most_common(12)
    print("LP %d x %d: %d steps; most common deltas:" % (m, n, tot), [(k, round(100.0 * v / tot, 1)) for k, v in top])
    for cand in ((1,), (1, 2), (1, 2, -2), (1, 2, -2, 3), (1, 2, -2, 3, -1, 4)):
        print("   candidates %s: %.1f %%" % (cand, 100.0 * sum(c[k] for k in cand) / tot))
    same_line = np.mean((e[1:] >> 4) == (e[:-1] >> 4))
    print("   next column in the same 16-column line: %.1f %%" % (100.0 * same_line))
    lp.close()
