"""Randomized GPU-vs-oracle cross-check of SIX::maxm / minm and Lineq::has_solution on systems WITH equalities
(convertEq2Ineq's quirks leave n/0 cells behind): run by hand, prints the first mismatches."""
import os, sys, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.six import SIX, has_solution
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0; total = 0; undefined = 0; bad_by_kind = {0: 0, 1: 0}
for kind in (1, 0):
    rng = np.random.default_rng(5150 + kind)
    six = SIX(ctx, kind)
    for it in range(N):
        nv = int(rng.integers(2, 7)); ml = int(rng.integers(1, 8)); me = int(rng.integers(1, 4))
        A = rng.integers(-3, 4, size=(ml, nv)); b = rng.integers(-4, 10, size=ml)
        xs = rng.integers(0, 4, size=nv)
        Ae = rng.integers(-2, 3, size=(me, nv)); be = Ae @ xs + (rng.integers(0, 2, size=me) if rng.random() < 0.25 else 0)
        c = rng.integers(-2, 6, size=nv)
        leq = np.concatenate([A, b[:, None]], axis=1).astype(np.int32); eq = np.concatenate([Ae, np.asarray(be).reshape(me, 1)], axis=1).astype(np.int32)
        tg = np.concatenate([c, [0]]).astype(np.int32); vc = gen.vc_nonneg(nv, False)
        if kind == 1: leq, eq, tg, vc = (gen.to_rat(x) for x in (leq, eq, tg, vc))
        else: leq, eq, tg, vc = (x.astype(np.float64) for x in (leq, eq, tg, vc))
        for is_max in (True, False):
            want = port.six_solve(kind, is_max, tg, vc, eq, leq)
            if want[0] == -7: undefined += 1; continue
            got = (six.maxm if is_max else six.minm)(tg, vc, eq, leq)
            total += 1
            same_v = np.array_equal(np.asarray(got[1]), np.asarray(want[1])) or (kind == 0 and np.asarray(got[1]).tobytes() == np.asarray(want[1]).tobytes())
            ok = got[0] == want[0] and same_v and (want[0] != 0 or np.array_equal(got[2], want[2]) or (kind == 0 and np.asarray(got[2]).tobytes() == np.asarray(want[2]).tobytes()))
            if not ok:
                bad += 1; bad_by_kind[kind] += 1
                if bad <= 5: print("MISMATCH kind", kind, "it", it, "max", is_max, "gpu", got[0], np.asarray(got[1]).tolist(), "oracle", want[0], np.asarray(want[1]).tolist(), "\n  leq", leq[..., 0].tolist() if kind else leq.tolist(), "\n  eq", eq[..., 0].tolist() if kind else eq.tolist(), "tg", tg[..., 0].tolist() if kind else tg.tolist())
        if kind == 1:
            for ii in (True, False):
                for uu in (True, False):
                    want = port.has_solution(leq, eq, vc, nv, ii, uu)
                    if want == -7: undefined += 1; continue
                    got = has_solution(ctx, leq, eq, vc, nv, ii, uu); total += 1
                    if got != want:
                        bad += 1
                        if bad <= 5: print("MISMATCH has_solution it", it, ii, uu, got, want)
print("compared", total, "mismatches", bad, "(fp64 %d, rational %d in SIX)" % (bad_by_kind[0], bad_by_kind[1]), "reference-undefined skipped", undefined)
