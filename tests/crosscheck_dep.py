"""GPU cross-check, larger than the collected tests' (run by hand: python tests/crosscheck_dep.py [seed] [count]):
xpg_dep_is_empty_batch_rat32 (reduce, feasibility objectives and both MIP walks on the device) against
DepPoly::is_empty composed from the CPU oracle's reduce + has_solution(int, unique), polyhedron by polyhedron."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.six import dep_is_empty_batch
from oracle.checker import Port
from tools import gen

ctx = xpoly_amd.Context(0)
port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bad = 0
for (rows, nv) in ((4, 2), (8, 3), (12, 4), (16, 6), (20, 5), (10, 8)):
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    mats[..., 1] = 1                                   # dependence polyhedra are integer systems
    empty, nodes = dep_is_empty_batch(ctx, mats)
    vc = gen.to_rat(gen.vc_nonneg(nv, False))
    hist = {}
    for b in range(nb):
        ok, res = port.reduce(mats[b], nv, True)
        if not ok:
            want = 1
        elif res.shape[0] == 0:
            want = 0
        else:
            h = port.has_solution(res, None, vc, nv, True, True)
            want = h if h < 0 else int(not h)
        hist[want] = hist.get(want, 0) + 1
        if empty[b] != want:
            bad += 1; print("mismatch", rows, nv, b, empty[b], want)
    print("%dx%d: %d polyhedra (%d nodes), verdicts %s, %d mismatches so far" % (rows, nv + 1, nb, nodes, hist, bad), flush=True)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
