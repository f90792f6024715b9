"""GPU cross-check (run by hand): the LDS batch kernel (xpg_six_batch_*) on LPs whose cells are not ordinary numbers --
fp64 inf / NaN, rational n/0, 0/d, negative and unreduced denominators -- against the oracle's SIX::maxm / minm."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != np.float64: return np.array_equal(a, b)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
for kind in (0, 1):
    bad = total = undefined = 0
    for it in range(40):
        m, n, nb = int(rng.integers(2, 9)), int(rng.integers(2, 8)), 16
        leqs, tgs = [], []
        for b_ in range(nb):
            A = rng.integers(-3, 6, size=(m, n)); A[rng.random((m, n)) < 0.3] = 0
            b = rng.integers(-2, 9, size=m); c = rng.integers(-2, 5, size=n)
            leq = np.concatenate([A, b[:, None]], axis=1); tg = np.concatenate([c, [0]])
            if kind == 0:
                leq = leq.astype(np.float64); tg = tg.astype(np.float64)
                for _ in range(int(rng.integers(1, 3))):
                    leq[int(rng.integers(0, m)), int(rng.integers(0, n + 1))] = rng.choice([np.inf, -np.inf, np.nan])
            else:
                leq = gen.to_rat(leq.astype(np.int32)); tg = gen.to_rat(tg.astype(np.int32))
                for _ in range(int(rng.integers(1, 4))):
                    i, j = int(rng.integers(0, m)), int(rng.integers(0, n + 1))
                    leq[i, j] = [(int(rng.integers(-3, 4)), 0), (0, int(rng.integers(2, 5))), (int(rng.integers(1, 5)), -int(rng.integers(1, 4))), (4, 6)][int(rng.integers(0, 4))]
            leqs.append(leq); tgs.append(tg)
        L = np.stack(leqs); T = np.stack(tgs)
        vc = gen.vc_nonneg(n, False); vc = vc.astype(np.float64) if kind == 0 else gen.to_rat(vc)
        for is_max in (True, False):
            st, v, sol = ctx.six_batch(kind, is_max, T, L)
            for b_ in range(nb):
                want = port.six_solve(kind, is_max, T[b_], vc, None, L[b_])
                if want[0] == -7: undefined += 1; continue
                total += 1
                ok = int(st[b_]) == want[0] and same(v[b_], want[1]) and (want[0] != 0 or same(sol[b_], want[2]))
                if not ok:
                    bad += 1
                    if bad <= 3: print("MISMATCH kind", kind, "it", it, "lp", b_, "max", is_max, "gpu", int(st[b_]), np.asarray(v[b_]).tolist(), "oracle", want[0], np.asarray(want[1]).tolist(), "\n leq", (L[b_].tolist()), "\n tg", T[b_].tolist())
    print("kind", kind, "compared", total, "mismatches", bad, "reference-undefined skipped", undefined)
