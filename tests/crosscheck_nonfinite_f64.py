"""GPU cross-check (run by hand): the HBM-resident fp64 loop (SIX::TwoStageMethod) on tableaux that hold inf / NaN cells --
what convertEq2Ineq's division by a zero entry leaves behind -- against the oracle: statuses at every iteration limit,
tableau, objective row and basis (NaNs compared as NaNs: x86 and the GPU give them different signs and payloads)."""
import os, sys, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import xpoly_amd
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
six = xpoly_amd.SIX(ctx, 0)
rng = np.random.default_rng(3)
bad = total = 0
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape: return False
    if a.dtype != np.float64: return a.tobytes() == b.tobytes()
    na, nb = np.isnan(a), np.isnan(b)                          # NaN payloads and signs differ between x86 and the GPU
    return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
for it in range(150):
    m, n = int(rng.integers(2, 8)), int(rng.integers(2, 7))
    A = rng.integers(-3, 6, size=(m, n)).astype(np.float64); A[rng.random((m, n)) < 0.3] = 0
    b = rng.integers(-2, 9, size=m).astype(np.float64); c = rng.integers(-2, 5, size=n).astype(np.float64)
    leq = np.concatenate([A, b[:, None]], axis=1); tg = np.concatenate([c, [0.0]])
    for _ in range(int(rng.integers(1, 3))):
        i, j = int(rng.integers(0, m)), int(rng.integers(0, n + 1))
        leq[i, j] = rng.choice([np.inf, -np.inf, np.nan])
    for K in (0, 1, 2, 3, 5, 1000):
        want = port.two_stage(0, leq, tg, K)
        if want["status"] == -7: continue
        six.set_param(0, K)
        got = six.TwoStageMethod(leq, tg)
        total += 1
        ok = got["status"] == want["status"] and (want["status"] == 2 or all(same(got[k], want[k]) for k in ("tab", "tgtf", "eq2bv")))
        if not ok:
            bad += 1
            if bad <= 3:
                print("MISMATCH it", it, "K", K, "gpu", got["status"], "oracle", want["status"], "\n leq", leq.tolist(), "tg", tg.tolist())
                if got["status"] == want["status"]:
                    for k in ("tab", "tgtf", "eq2bv"):
                        if not same(got[k], want[k]): print("   differs in", k, "\n   gpu", np.asarray(got[k]).tolist(), "\n   ora", np.asarray(want[k]).tolist())
            break
print("compared", total, "mismatching LPs", bad)
