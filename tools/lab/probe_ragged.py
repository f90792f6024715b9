"""GPU probe: the ragged dependence test (4096 polyhedra of 8 shapes) against per-shape uniform calls."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_ragged, ragged_pack_rat
from tools import gen
SHAPES = [(6, 2), (8, 3), (10, 4), (12, 4), (9, 5), (14, 5), (16, 6), (7, 3)]
ctx = xpoly_amd.Context(0)
rng = np.random.default_rng(5150)
polys = []
for k in range(512):
    rows, nv = SHAPES[int(rng.integers(0, len(SHAPES)))]
    m = gen.random_system(rng, rows, nv); m[..., 1] = 1
    polys.append(m)
polys = [polys[k % 512] for k in range(4096)]
packed = ragged_pack_rat(polys)
by_shape = {}
for p in polys:
    by_shape.setdefault(p.shape, []).append(p)
stacks = {k: np.ascontiguousarray(np.stack(v)) for k, v in by_shape.items()}
def t(f, n=5):
    f(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("ragged, one call (8 lanes)      %.2f ms" % t(lambda: dep_is_empty_ragged(ctx, packed=packed)))
print("per-shape uniform calls, serial %.2f ms" % t(lambda: [dep_is_empty_batch(ctx, s) for s in stacks.values()]))
for k, s in stacks.items():
    print("   shape %s x%d: %.2f ms" % (k[:2], len(s), t(lambda: dep_is_empty_batch(ctx, s), 3)))
uni = np.stack([gen.random_system(rng, 12, 4) for _ in range(256)]); uni[..., 1] = 1
uni = np.ascontiguousarray(np.tile(uni, (16, 1, 1, 1)))
print("uniform 4096 x (12 x 5)         %.2f ms" % t(lambda: dep_is_empty_batch(ctx, uni)))
big = [polys[k % 4096] for k in range(65536)]
pb = ragged_pack_rat(big)
print("ragged 65536 polyhedra of 8 shapes   %.2f ms" % t(lambda: dep_is_empty_ragged(ctx, packed=pb), 2))
uni16 = np.ascontiguousarray(np.tile(uni, (16, 1, 1, 1)))
print("uniform 65536 x (12 x 5)             %.2f ms" % t(lambda: dep_is_empty_batch(ctx, uni16), 2))
