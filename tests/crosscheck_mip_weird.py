"""GPU cross-check (run by hand): the MIP tree walks (xpg_mip_batch_*) and DepPoly::is_empty batches on problems whose
cells include n/0, 0/d, negative / unreduced denominators (rational) and inf / NaN (fp64), against the oracle."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.six import mip_batch
from oracle.checker import Port
from tools import gen
ctx = xpoly_amd.Context(0); port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
def weird(rng):
    return [(int(rng.integers(-3, 4)), 0), (0, int(rng.integers(2, 5))), (int(rng.integers(1, 5)), -int(rng.integers(1, 4))), (4, 6)][int(rng.integers(0, 4))]
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != np.float64: return np.array_equal(a, b)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
bad = total = undefined = 0
for kind in (1, 0):
    for it in range(24):
        m, nv, nb = int(rng.integers(1, 7)), int(rng.integers(2, 7)), 32
        is_bin = bool(rng.integers(0, 2))
        probs = [gen.random_mip(rng, m, nv, is_bin) for _ in range(nb)]
        rows = min(p["leq"].shape[0] for p in probs)
        L = np.stack([p["leq"][:rows] for p in probs]); T = np.stack([p["tgtf"] for p in probs])
        for b in range(nb):
            for _ in range(int(rng.integers(1, 3))):
                L[b, int(rng.integers(0, rows)), int(rng.integers(0, nv + 1))] = weird(rng)
        vc = gen.to_rat(gen.vc_nonneg(nv, False))
        if kind == 0:
            Lf = L[..., 0].astype(np.float64); Tf = T[..., 0].astype(np.float64)
            den0 = L[..., 1] == 0
            Lf[den0] = np.where(L[..., 0][den0] > 0, np.inf, np.where(L[..., 0][den0] < 0, -np.inf, np.nan))
            L, T, vc = Lf, Tf, vc[..., 0].astype(np.float64)
        for is_max in (True, False):
            st, v, sol, nodes = mip_batch(ctx, is_max, is_bin, T, L, kind=kind) if kind == 0 else mip_batch(ctx, is_max, is_bin, T, L)
            for b in range(nb):
                want = port.mip_solve(kind, is_max, is_bin, T[b], vc, None, L[b])
                if want[0] == -7: undefined += 1; continue
                total += 1
                ok = int(st[b]) == want[0] and same(v[b], want[1]) and (want[0] != 0 or same(sol[b], want[2]))
                if not ok:
                    bad += 1
                    if bad <= 4: print("MIP MISMATCH kind", kind, "bin", is_bin, "max", is_max, "gpu", int(st[b]), np.asarray(v[b]).tolist(), "oracle", want[0], np.asarray(want[1]).tolist(), "\n leq", L[b].tolist(), "tg", T[b].tolist())
    print("kind", kind, ": compared", total, "mismatches", bad, "undefined", undefined, flush=True)
from xpoly_amd.six import dep_is_empty_batch
for rows, nv in ((6, 3), (12, 4), (10, 6)):
    nb = 256
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    mats[..., 1] = 1
    for b in range(nb):
        mats[b, int(rng.integers(0, rows)), int(rng.integers(0, nv + 1))] = weird(rng)
    empty, nodes = dep_is_empty_batch(ctx, mats)
    vc = gen.to_rat(gen.vc_nonneg(nv, False))
    dbad = 0
    for b in range(nb):
        ok, res = port.reduce(mats[b], nv, True)
        if not ok: want = 1
        elif res.shape[0] == 0: want = 0
        else:
            h = port.has_solution(res, None, vc, nv, True, True)
            want = h if h < 0 else int(not h)
        if empty[b] != want:
            dbad += 1
            if dbad <= 3: print("dep mismatch", rows, nv, b, int(empty[b]), want, mats[b].tolist())
    bad += dbad
    print("dep_is_empty %dx%d: %d polyhedra, %d mismatches" % (rows, nv + 1, nb, dbad), flush=True)
print("TOTAL mismatches:", bad)
