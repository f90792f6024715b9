"""GPU cross-check (run by hand): the HBM-resident rational loop on tableaux that hold n/0 cells -- values Rational's
cross-multiplying comparisons do not order -- against the oracle's TwoStageMethod: small LPs through every path, and LPs
of more than 256 rows, where the pipelined loop's pick spans several workgroups and must hand such ratio tests to the
generic pick. Also SIX with equalities through the HBM path (XPG_FORCE_DEVICE_LP=1 in the environment)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from tools import gen
from oracle.checker import Port
RAT = 1
ctx = xpoly_amd.Context(0); port = Port()
six = xpoly_amd.SIX(ctx, RAT)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = total = skipped = 0
shapes = [(int(rng.integers(2, 12)), int(rng.integers(2, 10))) for _ in range(60)] + [(300, 40), (520, 30), (270, 90)]
for m, n in shapes:
    A = rng.integers(-3, 7, size=(m, n)).astype(np.int32); A[rng.random((m, n)) < 0.4] = 0
    b = (rng.integers(1, 4, size=m) * 6).astype(np.int32); c = rng.integers(-2, 5, size=n).astype(np.int32)
    leq = gen.to_rat(np.concatenate([A, b[:, None]], axis=1)); tg = gen.to_rat(np.concatenate([c, [0]]).astype(np.int32))
    for _ in range(max(2, m // 40)):                         # a few n/0 cells, positive and negative, also in the constant column
        i, j = int(rng.integers(0, m)), int(rng.integers(0, n + 1))
        leq[i, j] = (int(rng.choice([-2, -1, 1, 3])), 0)
    for K in (1, 3, 9, 40):
        want = port.two_stage(RAT, leq, tg, K)
        if want["status"] == -7:
            skipped += 1; continue
        six.set_param(0, K)
        got = six.TwoStageMethod(leq, tg)
        total += 1
        ok = got["status"] == want["status"] and (want["status"] == 2 or all(np.array_equal(got[k], want[k]) for k in ("tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv")))
        if not ok:
            bad += 1
            if bad <= 5: print("MISMATCH", m, n, "K", K, "gpu", got["status"], "oracle", want["status"])
print("compared", total, "mismatches", bad, "reference-undefined skipped", skipped)
