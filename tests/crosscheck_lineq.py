"""GPU cross-check, larger than the collected tests' (run by hand: python tests/crosscheck_lineq.py [seed] [count]): the row-elimination kernels against the CPU oracle on a few
thousand random systems per shape (both intersect modes, dark shadow, several eliminated variables, rank / det /
inv), including batches that mix systems with fractions not in lowest terms."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.lineq import Lineq
from oracle.checker import Port
from tools import gen

ctx = xpoly_amd.Context(0)
lq = Lineq(ctx)
port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 768
bad = 0


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return (a.shape[0] == 0 and b.shape[0] == 0) or (a.shape == b.shape and np.array_equal(a, b))


for rows, nv in ((5, 2), (16, 6), (17, 8), (30, 9), (40, 12), (64, 15), (70, 19)):
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    for b in range(0, nb, 5):                        # every fifth system: some k/k entries
        for _ in range(rows // 2):
            i, j, k = int(rng.integers(0, rows)), int(rng.integers(0, nv + 1)), int(rng.integers(2, 4))
            mats[b, i, j] = (mats[b, i, j, 0] * k, mats[b, i, j, 1] * k)
    for inter in (True, False):
        ok, res = lq.reduce(mats, nv, inter)
        for b in range(nb):
            wok, wres = port.reduce(mats[b], nv, inter)
            if ok[b] != wok or (wok and not same(res[b], wres)):
                bad += 1; print("reduce mismatch", rows, nv, inter, b)
    res = lq.removeIdenRow(mats)
    for b in range(nb):
        if not same(res[b], port.remove_iden_row(mats[b])):
            bad += 1; print("removeIdenRow mismatch", rows, nv, b)
    for u in sorted(set(int(x) for x in rng.integers(0, nv, size=2))):
        for dark in (False, True):
            ok, res = lq.fme(mats, nv, u, dark)
            for b in range(nb):
                wok, wres = port.fme(mats[b], nv, u, dark)
                if ok[b] != wok or not same(res[b], wres):
                    bad += 1; print("fme mismatch", rows, nv, u, dark, b)
    rk = lq.rank(mats)
    for b in range(nb):
        if rk[b] != port.rat_rank(mats[b]):
            bad += 1; print("rank mismatch", rows, nv, b)
    print("shape %dx%d: %d systems checked, %d mismatches so far" % (rows, nv + 1, nb, bad), flush=True)
for n in (3, 4, 7, 10, 16, 24):
    sq = np.stack([gen.random_square(rng, n) for _ in range(nb // 2)])
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(sq.shape[0]):
        wok, winv = port.rat_inv(sq[b])
        if rk[b] != port.rat_rank(sq[b]) or tuple(dt[b]) != port.rat_det(sq[b]) or ok[b] != wok or (wok and not np.array_equal(inv[b], winv)):
            bad += 1; print("gauss mismatch", n, b)
    print("square %d: %d matrices checked, %d mismatches so far" % (n, sq.shape[0], bad), flush=True)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
