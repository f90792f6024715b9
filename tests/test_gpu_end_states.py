"""Full-size solves pinned at their END against the REAL reference (tests/golden/g12_end_states.json, tools/gen_golden_end.py):
optimum detection, the basic solution, the in-order row sums of SIX::is_feasible (src/com/lpsol.h:784-822 -- what decides
status 0 against 3), calcFinalSolution (:1851-1899) and the objective SIX::maxm / minm return (:1993-2033, :1662-1732), at
sizes where every other fixture stops at SIX_TIME_OUT. Everything goes through the C ABI: xpg_six_maxm_f64 / xpg_six_minm_f64
with host arrays, and xpg_lp_two_stage on a device-resident LP.

Objectives are compared BIT FOR BIT (north_star allows 1e-9 relative on the float simplex objective; the replay is exact)."""
import json
import os
import zlib

import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
F64, RAT = 0, 1
NO_LIMIT = 0xFFFFFFFF
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_end_states.json")))


def checksum(a):
    a = np.ascontiguousarray(a)
    v = a.view(np.uint64).reshape(-1) if a.dtype.itemsize == 8 else a.view(np.uint32).reshape(-1).astype(np.uint64)
    return dict(crc32="%08x" % (zlib.crc32(a.tobytes()) & 0xFFFFFFFF), sum="%016x" % int(v.sum(dtype=np.uint64)),
                xor="%016x" % int(np.bitwise_xor.reduce(v)))


def val(kind, v):
    return float(v).hex() if kind == F64 else [int(v[0]), int(v[1])]


def check_two_stage_end(kind, lp, st, rec):
    assert st == rec["status"], (st, rec["status"])
    assert lp.pivots_done() == rec["pivots"]
    got = lp.read()
    assert list(got["tab"].shape[:2]) == rec["tab_shape"] and got["rhs"] == rec["rhs"]
    assert checksum(got["eq2bv"].astype(np.int32)) == rec["eq2bv"]
    assert [int(x) for x in got["eq2bv"][:32]] == rec["eq2bv_head"]
    assert checksum(got["bv2eq"].astype(np.int32)) == rec["bv2eq"]
    assert val(kind, got["tgtf"][got["rhs"]]) == rec["obj_const"]
    assert checksum(got["tgtf"]) == rec["tgtf"]
    assert checksum(got["tab"]) == rec["tab"]
    assert val(kind, got["maxv"]) == rec["maxv"]
    if rec["status"] == 0:
        assert checksum(got["sol"]) == rec["sol"]
    return got


def check_six(six, is_max, tg, vc, leq, rec, kind=F64):
    from xpoly_amd.six import six_last_profile
    six.set_param(0, NO_LIMIT)
    st, v, sol = (six.maxm if is_max else six.minm)(tg, vc, None, leq)
    pf = six_last_profile()
    assert pf["route"] == "HBM-resident loop", pf
    assert st == rec["status"], (st, rec["status"])
    assert val(kind, v) == rec["v"], (val(kind, v), rec["v"])
    if rec["status"] == 0:
        assert checksum(sol) == rec["sol"]
        assert int(np.count_nonzero(sol)) == rec["sol_nonzeros"]
    return pf


def test_bench_lp_to_its_natural_end_against_the_reference(ctx):
    """The LP bench.py times (gen.hard_lp_f64(4096, 4095), tableau 4096 x 8192) with NO iteration limit: the reference ends
    after 4165 pivots with SIX_OPTIMAL_IS_INFEASIBLE -- every structural variable basic, the pricing finds no positive cost,
    and a row sum of the 8191-term in-order feasibility check differs from its right-hand side. Status, pivot
    count (pinned to the reference: with max_iter = 4165 it stops at SIX_TIME_OUT in this state), whole tableau, objective
    row, basis."""
    import xpoly_amd
    rec = GOLD["bench_end"]
    assert rec["status"] == 3 and rec["pivots_pinned_by_reference"]
    leq, tgtf = gen.hard_lp_f64(4096, 4095)
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tgtf)
    st = lp.two_stage(NO_LIMIT)
    check_two_stage_end(F64, lp, st, rec)
    # ... and the same end from the iterate interface the bench loop uses: budget beyond the end
    base = lp.pivots_done()                              # (the counter runs over the handle's lifetime)
    lp.begin()
    assert lp.iterate(3840) == xpoly_amd.six.XPG_RUNNING
    assert lp.iterate(100000) == rec["status"]
    assert lp.pivots_done() - base == rec["pivots"]
    lp.close()


def test_large_lp_that_ends_succ_with_a_nonzero_optimum(ctx):
    """2309 x 2751 fp64 LP (tableau 2309 x 5061, 93 MB) that ends SIX_SUCC: gen.block_lp_f64 of the fixture's block seeds (half
    of the blocks integer data, half U(0.1, 1) data with real rounding). The k_rowcheck row sums must come out EQUAL in all
    2309 rows; optimum, solution and the whole end state are the reference's. Through xpg_lp_two_stage and through
    xpg_six_maxm_f64 (calcFinalSolution, the objective recomputed on the original tgtf)."""
    import xpoly_amd
    rec = GOLD["succ_two_stage"]
    assert rec["status"] == 0 and float.fromhex(rec["maxv"]) != 0.0
    leq, tgtf = gen.block_lp_f64(GOLD["succ_block_seeds"])
    assert list(leq.shape) == GOLD["succ_shape"] and leq.shape[0] >= 2048
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tgtf)
    st = lp.two_stage(NO_LIMIT)
    check_two_stage_end(F64, lp, st, rec)
    lp.close()
    six = xpoly_amd.SIX(ctx, F64)
    pf = check_six(six, True, tgtf, gen.vc_nonneg(leq.shape[1] - 1), leq, GOLD["succ_six_max"])
    assert pf["pivots"] == GOLD["succ_six_max"]["pivots"]


def test_bench_lp_through_six_maxm_to_the_end(ctx):
    """The bench LP once more, through the boundary a caller uses: ONE xpg_six_maxm_f64 call with vc = -I and no iteration
    limit must return SIX_OPTIMAL_IS_INFEASIBLE and v = 0 as SIX::maxm of the real reference does (lpsol.h:2024-2032)."""
    import xpoly_amd
    rec = GOLD["bench_six_max"]
    assert rec["status"] == 3
    leq, tg = gen.hard_lp_f64(4096, 4095)
    six = xpoly_amd.SIX(ctx, F64)
    pf = check_six(six, True, tg, gen.vc_nonneg(4095), leq, rec)
    assert pf["pivots"] == rec["pivots"] == GOLD["bench_end"]["pivots"]


def test_config_2_sized_lp_that_ends_succ_through_six_maxm(ctx):
    """At config 2's size (>= 4096 rows, >= 8192 variables: slack tableau 4099 x 15262, 500 MB) an LP that ENDS in SIX_SUCC with a
    non-zero optimum: gen.block_lp_f64(seeds, wide=True). One xpg_six_maxm_f64 call with host arrays (vc is 1 GB of which
    the diagonal is read): status, optimum bits, solution CRC and pivot count are the real reference's."""
    import xpoly_amd
    rec = GOLD["succ_big_six_max"]
    assert rec["status"] == 0 and float.fromhex(rec["v"]) != 0.0
    leq, tg = gen.block_lp_f64(GOLD["succ_big_block_seeds"], wide=True)
    assert list(leq.shape) == GOLD["succ_big_shape"] and leq.shape[0] >= 4096 and leq.shape[1] - 1 >= 8192
    six = xpoly_amd.SIX(ctx, F64)
    pf = check_six(six, True, tg, gen.vc_nonneg(leq.shape[1] - 1), leq, rec)
    assert pf["pivots"] == rec["pivots"]


def test_large_covering_lp_through_six_minm_to_the_end(ctx):
    """SIX::minm to a natural SIX_SUCC end on >= 2048 rows: gen.cover_lp_f64 (minimise c.x, A x >= b, x >= 0). The dual is built
    on the device (lpsol.h:1602-1629), solved to its optimum, the primal solution read off the dual's objective row
    (:1713-1716) and the optimum recomputed on the caller's tgtf (:1890-1898): status, optimum bits and solution CRC are
    the real reference's."""
    import xpoly_amd
    rec = GOLD["cover_six_min"]
    assert rec["status"] == 0 and float.fromhex(rec["v"]) != 0.0
    leq, tg = gen.cover_lp_f64(GOLD["cover_block_seeds"])
    assert list(leq.shape) == GOLD["cover_shape"] and leq.shape[0] >= 2048
    six = xpoly_amd.SIX(ctx, F64)
    check_six(six, False, tg, gen.vc_nonneg(leq.shape[1] - 1), leq, rec)


@pytest.mark.parametrize("which", ["dense_max_256x512", "dense_min_256x512"])
def test_dense_recipe_to_its_natural_end(ctx, which, monkeypatch):
    """The cfg-2b recipe of SURVEY 8d (gen.dense_lp_f64) at 256 x 512 with NO iteration limit, where the real reference can still
    reach its end: rounding leaves tiny positive costs, the loop goes on through the relaxed ratio pass, disableNV and
    findPivotNVandBVPair until the pivot-pair table is exhausted -- SIX_UNBOUND after 327 766 pivots (maxm),
    SIX_NO_PRI_FEASIBLE_SOL after phase 1's loop (minm). The HBM-resident loop must arrive at the same status after the same
    number of pivots. (At 4096 x 8192 the same end lies ~1e8 pivots away: months of the reference.)"""
    import xpoly_amd
    monkeypatch.setenv("XPG_FORCE_DEVICE_LP", "1")
    rec = GOLD[which]
    leq, tg = gen.dense_lp_f64(256, 512)
    six = xpoly_amd.SIX(ctx, F64)
    pf = check_six(six, which.startswith("dense_max"), tg, gen.vc_nonneg(512), leq, rec)
    if which.startswith("dense_max"):
        assert pf["pivots"] == rec["pivots"], (pf["pivots"], rec["pivots"])


def test_rational_cfg4_at_64_pivots_against_the_reference(ctx):
    """BASELINE configs[3]: the exact rational simplex on the 1024 x 2048 tableau after 64 pivots (the earlier fixtures stop at
    16): whole tableau, objective row, basis, bit for bit."""
    import xpoly_amd
    rec = GOLD["rational_k64"]
    leq, tgtf = gen.int_lp_rat(1024, 1023)
    lp = xpoly_amd.DeviceLP(ctx, RAT, leq, tgtf)
    st = lp.two_stage(rec["max_iter"])
    check_two_stage_end(RAT, lp, st, rec)
    lp.close()
