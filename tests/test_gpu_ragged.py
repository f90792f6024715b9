"""Ragged batches (include/xpoly_amd.h, "ragged batches"): problems of different shapes in one call, every answer equal
to the per-problem oracle's. The reference's workload is ragged -- DepPolyMgr::build emits polyhedra whose shape follows
statement depth and parameter count (src/eng/poly.cpp:1120-1224) and DepGraph::rebuild tests each (poly.cpp:268-314)."""
import time

import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
F64, RAT = 0, 1
SHAPES = [(6, 2), (8, 3), (10, 4), (12, 4), (9, 5), (14, 5), (16, 6), (7, 3)]      # (rows, variables): 8 mixed shapes


def dep_systems(rng, n):
    out = []
    for k in range(n):
        rows, nv = SHAPES[int(rng.integers(0, len(SHAPES)))]
        m = gen.random_system(rng, rows, nv)
        m[..., 1] = 1                                     # dependence polyhedra are integer systems
        out.append(m)
    return out


def test_dep_is_empty_4096_polyhedra_of_mixed_shapes(ctx, port):
    """4096 polyhedra of >= 6 mixed shapes in ONE call equal the per-polyhedron answers
    (the oracle composes reduce / has_solution as poly.cpp:530-573 does); the call takes no longer than 1.2 x a
    uniform call of as many polyhedra of the largest shape in the mix, and at most 0.7 x the per-shape calls back to back."""
    from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_ragged
    rng = np.random.default_rng(5150)
    uniq = dep_systems(rng, 512)
    polys = [uniq[k % 512] for k in range(4096)]
    assert len({p.shape for p in polys}) >= 6
    got, nodes = dep_is_empty_ragged(ctx, polys)
    want = {}
    for k in range(512):                                   # DepPoly::is_empty composed from the oracle's reduce + has_solution(int, unique)
        nv = uniq[k].shape[1] - 1
        ok, res = port.reduce(uniq[k], nv, True)
        if not ok:
            want[k] = 1
        elif res.shape[0] == 0:
            want[k] = 0
        else:
            h = port.has_solution(res, None, gen.to_rat(gen.vc_nonneg(nv, False)), nv, True, True)
            want[k] = h if h < 0 else int(not h)
    for k in range(4096):
        assert got[k] == want[k % 512], (k, polys[k].shape, got[k], want[k % 512])
    assert nodes > 0
    # throughput: beside the same polyhedra as one uniform call PER SHAPE back to back (what a caller had to do before),
    # and beside ONE uniform call of as many polyhedra of the largest shape in the mix (an upper bound on the work)
    from xpoly_amd.six import ragged_pack_rat
    packed = ragged_pack_rat(polys)
    by_shape = {}
    for p in polys:
        by_shape.setdefault(p.shape, []).append(p)
    stacks = [np.ascontiguousarray(np.stack(v)) for v in by_shape.values()]
    big = max(by_shape, key=lambda sh: sh[0] * sh[1])
    uni = np.ascontiguousarray(np.tile(np.stack(by_shape[big]), (4096 // len(by_shape[big]) + 1, 1, 1, 1))[:4096])

    def best_of(f, n=4):
        f()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        return min(ts)
    t_rag = best_of(lambda: dep_is_empty_ragged(ctx, packed=packed))
    t_ser = best_of(lambda: [dep_is_empty_batch(ctx, s) for s in stacks])
    t_uni = best_of(lambda: dep_is_empty_batch(ctx, uni))
    print("dep_is_empty, 4096 polyhedra of %d shapes: ragged call %.2f ms (%.0f k/s); per-shape uniform calls back to back "
          "%.2f ms; 4096 x the largest shape %s uniform %.2f ms" % (len(by_shape), t_rag * 1e3, 4.096 / t_rag, t_ser * 1e3, big[:2], t_uni * 1e3))
    assert t_rag <= 0.7 * t_ser, (t_rag, t_ser)
    assert t_rag <= 1.2 * t_uni, (t_rag, t_uni)


@pytest.mark.parametrize("kind", [F64, RAT])
def test_six_batch_ragged_matches_oracle(ctx, port, kind):
    from xpoly_amd.six import six_batch_ragged
    rng = np.random.default_rng(99 + kind)
    leqs, tgs = [], []
    for k in range(300):
        m, nv = int(rng.integers(1, 14)), int(rng.integers(1, 12))
        prob = gen.random_problem(rng, kind, int(rng.integers(0, 3)), m, nv, plain=True)
        leqs.append(prob["leq"]); tgs.append(prob["tgtf"])
    for is_max in (True, False):
        st, v, sol = six_batch_ragged(ctx, kind, is_max, tgs, leqs)
        for k in range(300):
            nv = leqs[k].shape[1] - 1
            w = port.six_solve(kind, is_max, tgs[k], gen.vc_nonneg(nv, kind == F64), None, leqs[k])
            assert st[k] == w[0], (k, is_max, st[k], w[0])
            if w[0] == 0:
                assert np.array_equal(np.asarray(v[k]), np.asarray(w[1])) and np.array_equal(sol[k], w[2]), (k, is_max)


def test_lineq_reduce_and_fme_ragged_match_oracle(ctx, port):
    from xpoly_amd.lineq import Lineq
    lq = Lineq(ctx)
    rng = np.random.default_rng(4242)
    mats, us = [], []
    for k in range(400):
        rows, nv = SHAPES[int(rng.integers(0, len(SHAPES)))]
        mats.append(gen.random_system(rng, rows, nv)); us.append(int(rng.integers(0, nv)))
    for inter in (True, False):
        ok, res = lq.reduce_ragged(mats, None, inter)
        for k in range(400):
            wok, wres = port.reduce(mats[k], mats[k].shape[1] - 1, inter)
            assert ok[k] == wok, (k, inter)
            if wok:
                assert res[k].shape == wres.shape and np.array_equal(res[k], wres), (k, inter)
    for dark in (False, True):
        ok, res = lq.fme_ragged(mats, us, None, dark)
        for k in range(400):
            wok, wres = port.fme(mats[k], mats[k].shape[1] - 1, us[k], dark)
            assert ok[k] == wok, (k, dark)
            assert (res[k].shape[0] == 0 and wres.shape[0] == 0) or (res[k].shape == wres.shape and np.array_equal(res[k], wres)), (k, dark)
    # the same through the sizing call (what batches whose worst case exceeds 256 MB take): XPG_ERR_SHAPE with the offsets
    # filled, then a buffer of exactly that size
    ok1, res1 = lq.fme_ragged(mats, us, None, False)
    lq.RAGGED_FME_ONE_CALL_BYTES = 0
    try:
        ok2, res2 = lq.fme_ragged(mats, us, None, False)
    finally:
        del lq.RAGGED_FME_ONE_CALL_BYTES
    assert np.array_equal(ok1, ok2) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(res1, res2))


def test_ragged_and_packed_error_paths(ctx):
    """Shapes the reference would only ASSERT on come back as XPG_ERR_SHAPE; a device slot too small for a result is
    XPG_ERR_UNSUPPORTED with the rows needed; empty batches are no-ops."""
    import ctypes as C
    from xpoly_amd._capi import lib, vp
    from xpoly_amd.six import dep_is_empty_ragged, ragged_pack_rat
    rng = np.random.default_rng(3)
    mats = [gen.random_system(rng, 6, 3), gen.random_system(rng, 9, 4)]
    flat, rows, cols, off = ragged_pack_rat(mats)
    out = np.zeros(2, dtype=np.int32)
    # a polyhedron without rows is EMPTY for DepPoly::is_empty (poly.cpp:533-535): answered, not an error, and the other
    # systems of the call get their own answers
    want, _ = dep_is_empty_ragged(ctx, mats)
    no_rows = rows.copy(); no_rows[1] = 0
    out[:] = -9
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(2), vp(flat), vp(no_rows), vp(cols), vp(off), vp(out), None) == 0
    assert out[1] == 1 and out[0] == want[0]
    no_rows[0] = 0
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(2), None, vp(no_rows), vp(cols), vp(off), vp(out), None) == 0
    assert out.tolist() == [1, 1]
    bad_rows = rows.copy(); bad_rows[1] = -1
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(2), vp(flat), vp(bad_rows), vp(cols), vp(off), vp(out), None) == -3
    bad_cols = cols.copy(); bad_cols[0] = 1
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(2), vp(flat), vp(rows), vp(bad_cols), vp(off), vp(out), None) == -3
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(0), vp(flat), vp(rows), vp(cols), vp(off), vp(out), None) == 0
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(0), None, None, None, None, None, None) == 0     # an empty SCoP
    assert lib().xpg_dep_is_empty_batch_ragged_rat32(ctx._h, C.c_int(2), None, vp(rows), vp(cols), vp(off), vp(out), None) == -3
    # packed fme: a device slot of 2 rows cannot hold the result of a 10-row system with positive and negative rows
    m = np.ascontiguousarray(np.stack([gen.random_system(rng, 10, 4) for _ in range(4)]))
    offs = np.zeros(5, dtype=np.int64); ok = np.zeros(4, dtype=np.int32); view = C.c_void_p()
    rc = lib().xpg_lineq_fme_batch_packed_rat32(ctx._h, C.c_int(4), vp(m), C.c_int(10), C.c_int(5), C.c_int(4), C.c_int(0), C.c_int(0),
                                                C.c_int(10), None, C.c_longlong(0), C.byref(view), vp(offs), vp(ok))
    assert rc in (0, -4)
    if rc == -4:
        assert offs[0] < 0                                  # -(rows the largest result needs)
    rc = lib().xpg_lineq_fme_batch_packed_rat32(ctx._h, C.c_int(0), vp(m), C.c_int(10), C.c_int(5), C.c_int(4), C.c_int(0), C.c_int(0),
                                                C.c_int(0), None, C.c_longlong(0), C.byref(view), vp(offs), vp(ok))
    assert rc == 0 and offs[0] == 0
    # and the handle still works
    got, _ = dep_is_empty_ragged(ctx, mats)
    assert got.shape == (2,)
