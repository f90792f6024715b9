"""GPU: the warm-started branch and bound (opt-in, NON-parity; xpg_mip_warm_f64). Every node
is re-optimised from its parent's final tableau by the dual simplex. There is no reference behaviour to match (the
reference solves every node from scratch and its depth-first walk depends on a fork counter), so the check is the
mathematics: the optimum equals scipy's HiGHS milp, the returned point is feasible and integral, and re-optimising
a node costs a few dual pivots where a cold solve costs the root's primal count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _milp(c, A, b, is_max):
    from scipy.optimize import Bounds, LinearConstraint, milp
    r = milp(c=-c if is_max else c, constraints=LinearConstraint(A, -np.inf, b), integrality=np.ones(len(c)),
             bounds=Bounds(0, np.inf))
    return r


@pytest.mark.parametrize("seed", range(12))
def test_warm_mip_reaches_the_integer_optimum(ctx, seed):
    from xpoly_amd.six import mip_warm
    rng = np.random.default_rng(1000 + seed)
    nv, m = int(rng.integers(4, 13)), int(rng.integers(2, 7))
    is_bin = seed % 3 == 0
    A = rng.integers(1, 10, size=(m, nv)).astype(np.float64)
    b = np.floor(A.sum(axis=1) * rng.uniform(0.3, 0.6, size=m)) + 0.5 * (seed % 2)
    c = rng.integers(1, 12, size=nv).astype(np.float64)
    if is_bin:                                          # 0-1 bounds are rows of the problem, as for the parity MIP
        A = np.concatenate([A, np.eye(nv)], axis=0); b = np.concatenate([b, np.ones(nv)])
    leq = np.concatenate([A, b[:, None]], axis=1)
    tgtf = np.concatenate([c, [0.0]])
    ref = _milp(c, A, b, True)
    assert ref.status == 0
    st, v, sol, stats = mip_warm(ctx, True, tgtf, leq, is_bin)
    assert st == 0, (st, stats)
    assert abs(v - (-ref.fun)) <= 1e-7 * max(1.0, abs(ref.fun)), (v, -ref.fun, stats)
    x = sol[:nv]
    assert np.abs(x - np.round(x)).max() <= 1e-6 and (x >= -1e-9).all()
    assert (A @ x <= b + 1e-7).all() and abs(c @ x - v) <= 1e-7 * max(1.0, abs(v))
    assert stats["nodes"] >= 1
    if stats["nodes"] > 4:                              # a warm node needs a handful of dual pivots, not a fresh solve
        assert stats["dual_pivots"] / (stats["nodes"] - 1) < max(6.0, 0.75 * stats["root_pivots"]), stats


def test_warm_mip_minimise_and_infeasible(ctx):
    from xpoly_amd.six import mip_warm
    # minimise x + 2y subject to -x - y <= -3.5 (x + y >= 3.5), x, y >= 0 integer: 4 at (4, 0)
    leq = np.array([[-1.0, -1.0, -3.5]]); tgtf = np.array([1.0, 2.0, 0.0])
    st, v, sol, stats = mip_warm(ctx, False, tgtf, leq)
    assert st == 0 and abs(v - 4.0) < 1e-9 and np.allclose(sol[:2], [4.0, 0.0])
    # 2x = 1 has no integer point: 2x <= 1, -2x <= -1
    leq = np.array([[2.0, 1.0], [-2.0, -1.0]]); tgtf = np.array([1.0, 0.0])
    st, v, sol, stats = mip_warm(ctx, True, tgtf, leq)
    assert st == 2


def test_warm_mip_batch_reaches_every_integer_optimum(ctx):
    """The batch form (xpg_mip_warm_batch_f64): 1024 knapsacks of the MIP leg's shape (24 0-1 variables, two capacity
    rows + the x_j <= 1 rows) in ONE launch, a tree per workgroup -- every optimum equals scipy's HiGHS milp, every point
    is feasible and integral, and the answers equal the one-tree form's on a sample; then random integer programs with
    negative right-hand sides (the root needs its phase one), minimisation, and an infeasible program in the batch."""
    from tools import gen
    from xpoly_amd.six import mip_warm, mip_warm_batch
    nb, nv = 1024, 24
    leq_r, tg_r = gen.knapsack_batch_rat(nb, nv)
    leq = leq_r[..., 0].astype(np.float64); tg = tg_r[..., 0].astype(np.float64)
    st, v, sol, stats = mip_warm_batch(ctx, True, tg, leq, is_bin=True)
    assert (st == 0).all(), np.bincount(st.clip(-10) + 10)
    assert stats["nodes"] >= nb and stats["max_depth"] <= nv
    for b in range(0, nb, 8):
        A, rhs, c = leq[b, :, :nv], leq[b, :, nv], tg[b, :nv]
        ref = _milp(c, A, rhs, True)
        assert ref.status == 0 and abs(v[b] - (-ref.fun)) <= 1e-7 * max(1.0, abs(ref.fun)), (b, v[b], -ref.fun)
    x = sol[:, :nv]
    assert np.abs(x - np.round(x)).max() <= 1e-6 and (x >= -1e-9).all()
    assert (np.einsum("bij,bj->bi", leq[:, :, :nv], x) <= leq[:, :, nv] + 1e-7).all()
    assert np.abs(np.einsum("bj,bj->b", tg[:, :nv], x) - v).max() <= 1e-6
    for b in (0, 17, 512, 1023):
        s1, v1, _, _ = mip_warm(ctx, True, tg[b], leq[b], True)
        assert s1 == st[b] and abs(v1 - v[b]) <= 1e-9 * max(1.0, abs(v1))
    if stats["nodes"] > 4 * nb:
        assert stats["dual_pivots"] / (stats["nodes"] - nb) < 12.0, stats
    # general integers, some rows with negative constants (x_0 + x_1 >= 2 as -x_0 - x_1 <= -2), both senses
    rng = np.random.default_rng(77)
    nb2, nv2, m2 = 96, 7, 5
    A = rng.integers(1, 9, size=(nb2, m2, nv2)).astype(np.float64)
    rhs = np.floor(A.sum(axis=2) * rng.uniform(0.6, 1.4, size=(nb2, m2))) + 0.5
    A[:, 0, :] = 0.0; A[:, 0, 0] = -1.0; A[:, 0, 1] = -1.0; rhs[:, 0] = -2.0
    c = rng.integers(1, 12, size=(nb2, nv2)).astype(np.float64)
    A[5, 1, :] = 0.0; A[5, 1, 0] = 2.0; rhs[5, 1] = 1.0; A[5, 2, :] = 0.0; A[5, 2, 0] = -2.0; rhs[5, 2] = -1.0     # 2 x_0 = 1: no integer point
    leq2 = np.concatenate([A, rhs[:, :, None]], axis=2); tg2 = np.concatenate([c, np.zeros((nb2, 1))], axis=1)
    for is_max in (True, False):
        st2, v2, sol2, _ = mip_warm_batch(ctx, is_max, tg2, leq2)
        for b in range(nb2):
            ref = _milp(c[b], A[b], rhs[b], is_max)
            if ref.status != 0:
                assert st2[b] == 2, (b, is_max, st2[b], ref.status)
                continue
            want = -ref.fun if is_max else ref.fun
            assert st2[b] == 0 and abs(v2[b] - want) <= 1e-7 * max(1.0, abs(want)), (b, is_max, st2[b], v2[b], want)
            xb = sol2[b, :nv2]
            assert np.abs(xb - np.round(xb)).max() <= 1e-6 and (A[b] @ xb <= rhs[b] + 1e-7).all()
        assert st2[5] == 2
