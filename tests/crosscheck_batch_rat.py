"""GPU cross-check, larger than the collected tests' (run by hand: python tests/crosscheck_batch_rat.py [seed] [count]): batched rational LPs (k_batch<R32>, canonical and generic
arithmetic mixed in one launch) against the CPU oracle -- status, objective, solution -- maxm and minm."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from oracle.checker import Port
from tools import gen

RAT = xpoly_amd.RAT
ctx = xpoly_amd.Context(0)
port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
bad = 0
for (m, nv, fam) in ((3, 4, 0), (5, 3, 1), (8, 8, 2), (6, 11, 1), (12, 7, 0), (12, 16, 1), (20, 30, 1), (32, 40, 1), (26, 24, 0)):
    probs = [gen.random_problem(rng, RAT, fam, m, nv, plain=True) for _ in range(nb)]
    leq = np.stack([p["leq"] for p in probs]); tg = np.stack([p["tgtf"] for p in probs])
    for b in range(0, nb, 4):                        # every fourth LP: a few k/k entries (not in lowest terms)
        for _ in range(3):
            i, j, k = int(rng.integers(0, m)), int(rng.integers(0, nv + 1)), int(rng.integers(2, 4))
            leq[b, i, j] = (leq[b, i, j, 0] * k, leq[b, i, j, 1] * k)
    for is_max in (True, False):
        status, v, sol = ctx.six_batch(RAT, is_max, tg, leq)
        for b in range(nb):
            want = port.six_solve(RAT, is_max, tg[b], probs[b]["vc"], None, leq[b])
            if status[b] != want[0] or not np.array_equal(np.asarray(v[b]).ravel(), np.asarray(want[1]).ravel()) or \
               (want[0] == 0 and not np.array_equal(sol[b], want[2])):
                bad += 1; print("mismatch", m, nv, fam, is_max, b, status[b], want[0])
    print("shape %dx%d family %d: %d LPs x 2 checked, %d mismatches so far" % (m, nv + 1, fam, nb, bad), flush=True)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
