"""GPU: the opt-in NON-PARITY modes of the device-resident fp64 loop (SURVEY section 8f, N4;
xpg_lp_set_options): Dantzig pricing and a tolerant is_feasible. There is no reference behaviour to
match here, so the check is the mathematics: the optimum agrees with scipy's HiGHS to 1e-7 relative
and the returned point is feasible. The parity mode of the same problems is untouched (and reports
most fp64 optima as SIX_OPTIMAL_IS_INFEASIBLE, which is the reference's own behaviour)."""
import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
F64, RAT = 0, 1


@pytest.mark.parametrize("m,n", [(12, 9), (40, 39), (96, 80), (256, 200)])
def test_dantzig_tolerant_mode_reaches_the_optimum(ctx, m, n):
    import xpoly_amd
    from scipy.optimize import linprog
    leq, tg = gen.hard_lp_f64(m, n)
    A, b, c = leq[:, :-1], leq[:, -1], tg[:-1]
    ref = linprog(-c, A_ub=A, b_ub=b, bounds=[(0, None)] * n, method="highs")
    assert ref.status == 0
    want = -ref.fun + tg[-1]

    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    st_parity = lp.two_stage()
    piv_parity = lp.pivots_done()
    lp.close()

    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    lp.set_options(pricing=1, feas_rel_tol=1e-9)
    st = lp.two_stage()
    piv = lp.pivots_done()
    out = lp.read(want_tab=False)
    lp.close()
    assert st == 0, (st, st_parity)
    assert abs(out["maxv"] - want) <= 1e-7 * max(1.0, abs(want)), (out["maxv"], want)
    x = out["sol"][:n]
    assert (x >= -1e-9).all() and (A @ x <= b + 1e-7 * np.maximum(1.0, np.abs(b))).all()
    assert abs(c @ x + tg[-1] - want) <= 1e-7 * max(1.0, abs(want))
    # the reference's rule on the same LP: optimum (exact, or "infeasible" by its 1e-17 '=='), or
    # SIX_UNBOUND once its anti-cycling pair table has nothing left to offer -- all bug-compatible
    assert st_parity in (0, 1, 3)
    if m >= 96:
        assert piv < piv_parity, (piv, piv_parity)  # largest-coefficient pricing needs fewer pivots here
    print("m=%d n=%d: pivots parity %d -> dantzig %d" % (m, n, piv_parity, piv))


@pytest.mark.parametrize("B", [4, 16])
def test_dantzig_mode_through_the_blocked_loop(B, monkeypatch):
    """The same non-parity mode through the blocked loop (Dantzig look-ahead as per-workgroup keys),
    forced onto a small LP: same optimum as HiGHS, and the same pivot count as the pipelined loop."""
    import xpoly_amd
    from scipy.optimize import linprog
    m, n = 96, 80
    leq, tg = gen.hard_lp_f64(m, n)
    A, b, c = leq[:, :-1], leq[:, -1], tg[:-1]
    ref = linprog(-c, A_ub=A, b_ub=b, bounds=[(0, None)] * n, method="highs")
    want = -ref.fun + tg[-1]
    res = {}
    for mode in ("pipe", "block"):
        monkeypatch.setenv("XPG_LOOP", mode)
        monkeypatch.setenv("XPG_BLOCK", str(B))
        c2 = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c2, F64, leq, tg)
        lp.set_options(pricing=1, feas_rel_tol=1e-9)
        st = lp.two_stage()
        res[mode] = (st, lp.pivots_done(), lp.read(want_tab=True), lp.trace().copy())
        lp.close(); c2.close()
    for mode in res:
        assert res[mode][0] == 0
        assert abs(res[mode][2]["maxv"] - want) <= 1e-7 * max(1.0, abs(want))
    assert res["pipe"][1] == res["block"][1] and np.array_equal(res["pipe"][3], res["block"][3])
    assert np.array_equal(res["pipe"][2]["tab"].view(np.uint64), res["block"][2]["tab"].view(np.uint64))


def test_options_are_fp64_only_and_reversible(ctx, port):
    import xpoly_amd
    rng = np.random.default_rng(3)
    p = gen.random_problem(rng, RAT, 1, 6, 5, plain=True)
    lp = xpoly_amd.DeviceLP(ctx, RAT, p["leq"], p["tgtf"])
    with pytest.raises(xpoly_amd._capi.XpgError):
        lp.set_options(pricing=1)
    lp.set_options(0, 0.0)                           # the neutral setting is accepted everywhere
    lp.close()
    # set and reset on an fp64 handle: the parity result comes back bit for bit
    leq, tg = gen.hard_lp_f64(24, 23)
    want = port.two_stage(F64, leq, tg, 0xFFFFFFFF)
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    lp.set_options(1, 1e-9)
    lp.set_options(0, 0.0)
    assert lp.two_stage() == want["status"]
    got = lp.read()
    lp.close()
    assert np.array_equal(got["tab"].view(np.uint64), want["tab"].view(np.uint64))
