"""GPU cross-check (run by hand): the row-elimination kernels on systems whose cells include what Rational(INT, INT)
stores as given -- n/0, 0/d, negative and unreduced denominators -- against the oracle (reduce both modes,
removeIdenRow, fme with and without the dark shadow, rank; det / inv on square matrices)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.lineq import Lineq
from oracle.checker import Port
from tools import gen
ctx = xpoly_amd.Context(0); lq = Lineq(ctx); port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bad = 0
def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return (a.shape[0] == 0 and b.shape[0] == 0) or (a.shape == b.shape and np.array_equal(a, b))
def weird(rng):
    return [(int(rng.integers(-3, 4)), 0), (0, int(rng.integers(2, 5))), (int(rng.integers(1, 5)), -int(rng.integers(1, 4))), (4, 6), (0, -1)][int(rng.integers(0, 5))]
for rows, nv in ((5, 2), (12, 4), (16, 6), (30, 9)):
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    for b in range(nb):
        for _ in range(int(rng.integers(1, 4))):
            mats[b, int(rng.integers(0, rows)), int(rng.integers(0, nv + 1))] = weird(rng)
    for inter in (True, False):
        ok, res = lq.reduce(mats, nv, inter)
        for b in range(nb):
            wok, wres = port.reduce(mats[b], nv, inter)
            if ok[b] != wok or (wok and not same(res[b], wres)):
                bad += 1
                if bad <= 5: print("reduce mismatch", rows, nv, inter, b, mats[b].tolist())
    res = lq.removeIdenRow(mats)
    for b in range(nb):
        if not same(res[b], port.remove_iden_row(mats[b])):
            bad += 1
            if bad <= 5: print("removeIdenRow mismatch", rows, nv, b)
    for u in sorted(set(int(x) for x in rng.integers(0, nv, size=2))):
        for dark in (False, True):
            ok, res = lq.fme(mats, nv, u, dark)
            for b in range(nb):
                wok, wres = port.fme(mats[b], nv, u, dark)
                if ok[b] != wok or not same(res[b], wres):
                    bad += 1
                    if bad <= 5: print("fme mismatch", rows, nv, u, dark, b, mats[b].tolist())
    rk = lq.rank(mats)
    for b in range(nb):
        if rk[b] != port.rat_rank(mats[b]):
            bad += 1
            if bad <= 5: print("rank mismatch", rows, nv, b, mats[b].tolist())
    for unit in (False, True):
        rk2, bas = lq.rankBasis(mats, unit)
        for b in range(nb):
            wr, wb = port.rat_rank_basis(mats[b], unit)
            if rk2[b] != wr or not same(bas[b], wb):
                bad += 1
                if bad <= 5: print("rankBasis mismatch", rows, nv, unit, b, mats[b].tolist())
    nl = lq.null(mats)
    for b in range(nb):
        if not np.array_equal(nl[b], port.rat_null(mats[b])):
            bad += 1
            if bad <= 5: print("null mismatch", rows, nv, b, mats[b].tolist())
    mv = lq.move2var(np.concatenate([mats, mats[:, :, :2]], axis=2), nv, nv + 1, nv + 2)
    for b in range(0, nb, 8):
        if not np.array_equal(mv[b], port.move2var(np.concatenate([mats[b], mats[b][:, :2]], axis=1), nv, nv + 1, nv + 2)):
            bad += 1
            if bad <= 5: print("move2var mismatch", rows, nv, b)
    if rows <= 12:
        okb, bnd = lq.calcBound(mats[:64], nv, cap_rows=4 * rows * rows)
        for b in range(64):
            try:
                wok, wb = port.calc_bound(mats[b], nv)
            except RuntimeError:
                continue
            gok, gb = okb[b], bnd[b]
            try:
                while gok < 0:                                 # an intermediate system outgrew the slots: -rows needed (of the step that overflowed); ask again
                    o2, b2 = lq.calcBound(mats[b:b + 1], nv, cap_rows=2 * int(-gok) + 64)
                    gok, gb = o2[0], b2[0]
            except Exception:                                  # (beyond what one call can hold: not a parity question)
                continue
            if gok != wok or (wok and any(not same(gb[j], wb[j]) for j in range(nv))):
                bad += 1
                if bad <= 5: print("calcBound mismatch", rows, nv, b, mats[b].tolist())
    print("shape %dx%d: %d systems checked, %d mismatches so far" % (rows, nv + 1, nb, bad), flush=True)
for n in (3, 4, 7):
    sq = np.stack([gen.random_square(rng, n) for _ in range(nb // 2)])
    for b in range(sq.shape[0]):
        sq[b, int(rng.integers(0, n)), int(rng.integers(0, n))] = weird(rng)
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(sq.shape[0]):
        wok, winv = port.rat_inv(sq[b])
        if rk[b] != port.rat_rank(sq[b]) or tuple(dt[b]) != port.rat_det(sq[b]) or ok[b] != wok or (wok and not np.array_equal(inv[b], winv)):
            bad += 1
            if bad <= 8: print("gauss mismatch", n, b, "rank", rk[b], port.rat_rank(sq[b]), "det", tuple(dt[b]), port.rat_det(sq[b]), "inv ok", ok[b], wok, sq[b].tolist())
    print("square %d: %d matrices checked, %d mismatches so far" % (n, sq.shape[0], bad), flush=True)
print("TOTAL mismatches:", bad)
