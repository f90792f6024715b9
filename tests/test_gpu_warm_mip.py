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
