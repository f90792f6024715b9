import os, sys, numpy as np, collections
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.six import SIX
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
rng = np.random.default_rng(5150 + 0)
six = SIX(ctx, 0)
hist = collections.Counter(); shown = 0
def fold_finite(leq, eq):
    # convertEq2Ineq in float arithmetic (lpsol.h:1197-1278): does it leave inf / nan behind?
    L = leq.copy(); E = eq.copy(); rhs = L.shape[1] - 1
    used = [False] * len(E)
    with np.errstate(all="ignore"):
        for j in range(rhs):
            hits = [i for i in range(len(E)) if not used[i] and E[i, j] != 0]
            if len(hits) != 1: continue
            at = hits[0]; used[at] = True
            for q in range(len(L)):
                coef = L[q, j]
                if coef == 0: continue
                if q >= E.shape[1]: return None
                t = E[at].copy(); lead = t[q]
                if lead != 1: t = t * (1.0 / lead)
                if coef != 1: t = t * coef if coef != 0 else t * 0
                L[q, j] = 0; t[rhs:] = -t[rhs:]; L[q] = t + L[q]
    return bool(np.isfinite(L).all())
for it in range(300):
    nv = int(rng.integers(2, 7)); ml = int(rng.integers(1, 8)); me = int(rng.integers(1, 4))
    A = rng.integers(-3, 4, size=(ml, nv)); b = rng.integers(-4, 10, size=ml)
    xs = rng.integers(0, 4, size=nv)
    Ae = rng.integers(-2, 3, size=(me, nv)); be = Ae @ xs + (rng.integers(0, 2, size=me) if rng.random() < 0.25 else 0)
    c = rng.integers(-2, 6, size=nv)
    leq = np.concatenate([A, b[:, None]], axis=1).astype(np.float64); eq = np.concatenate([Ae, np.asarray(be).reshape(me, 1)], axis=1).astype(np.float64)
    tg = np.concatenate([c, [0]]).astype(np.float64); vc = gen.vc_nonneg(nv, False).astype(np.float64)
    for is_max in (True, False):
        want = port.six_solve(0, is_max, tg, vc, eq, leq)
        if want[0] == -7: continue
        got = (six.maxm if is_max else six.minm)(tg, vc, eq, leq)
        same = got[0] == want[0] and np.asarray(got[1]).tobytes() == np.asarray(want[1]).tobytes()
        if not same:
            fin = fold_finite(leq, eq)
            hist[(is_max, got[0], want[0], fin)] += 1
            if shown < 2:
                shown += 1
                print("finite-fold mismatch it", it, "max", is_max, "gpu", got[0], float(np.asarray(got[1])), "oracle", want[0], float(np.asarray(want[1])), "\n leq", leq.tolist(), "\n eq", eq.tolist(), "\n tg", tg.tolist())
print("mismatch histogram (is_max, gpu status, oracle status, folded system finite):")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]): print("  ", k, v)
