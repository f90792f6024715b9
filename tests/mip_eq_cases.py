"""Random MIPs with equalities at the root (the shape PolyTran::FeaSchedule hands to MIP::maxm / minm,
src/eng/poly.cpp:5118-5130: x >= 0, equalities `sys`, inequalities from reviseTargetFunc) compared with the oracle's
restatement of MIP::RecusivePart + SIX::convertEq2Ineq. Used in-process (the device tree walk) and in a process
started with XPG_MIP_DEVICE=0 (the host controller)."""
import numpy as np

from tools import gen

F64, RAT = 0, 1


def random_mip_eq(rng, m_leq, m_eq, nv, is_bin):
    hi = 2 if is_bin else 5
    xs = rng.integers(0, hi, size=nv)                          # a point the equalities pass through
    Ae = rng.integers(-2, 4, size=(m_eq, nv))
    Ae *= rng.random((m_eq, nv)) < 0.7
    be = Ae @ xs
    if rng.random() < 0.2:
        be = be + rng.integers(0, 2, size=m_eq)                # sometimes off the lattice / infeasible
    A = rng.integers(0, 6, size=(m_leq, nv))
    if rng.random() < 0.3:
        A = A - rng.integers(0, 3, size=(m_leq, nv))
    b = A @ xs + rng.integers(0, 2 * nv + 2, size=m_leq)
    c = rng.integers(-2, 8, size=nv)
    eq = np.concatenate([Ae, be[:, None]], axis=1).astype(np.int32)
    leq = np.concatenate([A, b[:, None]], axis=1).astype(np.int32)
    if is_bin and rng.random() < 0.6:
        ub = np.zeros((nv, nv + 1), dtype=np.int32)
        ub[np.arange(nv), np.arange(nv)] = 1
        ub[:, nv] = 1
        leq = np.concatenate([leq, ub], axis=0)
    tgtf = np.concatenate([c, [int(rng.integers(0, 3))]]).astype(np.int32)
    prob = dict(tgtf=gen.to_rat(tgtf), vc=gen.to_rat(gen.vc_nonneg(nv, False)), eq=gen.to_rat(eq),
                leq=gen.to_rat(leq) if leq.shape[0] else None)
    if rng.random() < 0.15:
        prob["ind"] = (rng.random(nv + 1) < 0.3).astype(np.uint8)
    return prob


def run(ctx, port, kind, seed, count):
    """Returns (compared, status histogram); raises AssertionError on the first difference."""
    from xpoly_amd.six import MIP
    rng = np.random.default_rng(seed)
    mip = MIP(ctx, kind)
    compared, seen = 0, {}
    for it in range(count):
        m_leq, m_eq, nv = int(rng.integers(0, 6)), int(rng.integers(1, 4)), int(rng.integers(2, 7))
        is_bin = bool(rng.integers(0, 2))
        p = random_mip_eq(rng, m_leq, m_eq, nv, is_bin)
        if kind == F64:
            p = {k: (v[..., 0].astype(np.float64) if (k != "ind" and v is not None) else v) for k, v in p.items()}
        for is_max in (True, False):
            want = port.mip_solve(kind, is_max, is_bin, p["tgtf"], p["vc"], p["eq"], p["leq"], p.get("ind"))
            if want[0] == -7:
                continue                                       # the reference's behaviour is undefined there
            got = (mip.maxm if is_max else mip.minm)(p["tgtf"], p["vc"], p["eq"], p["leq"], is_bin, p.get("ind"))
            assert got[0] == want[0], (it, is_max, got[0], want[0])
            assert np.array_equal(np.asarray(got[1]), np.asarray(want[1])), (it, is_max, got[1], want[1])
            if want[0] == 0:
                assert np.array_equal(got[2], want[2]), (it, is_max)
            compared += 1
            seen[int(want[0])] = seen.get(int(want[0]), 0) + 1
    return compared, seen
