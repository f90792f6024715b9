"""CPU suite, part 1: pins the oracle (our CPU restatement, oracle/oracle.cpp) to the
golden vectors generated from the real reference by tools/gen_golden.py. If the
reference build is present (authoring container), also re-checks live against it.
"""
import json
import os

import numpy as np
import pytest

from tools import gen

F64, RAT = 0, 1
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dec(lst, kind, shape=None):
    if kind == F64:
        a = np.array([float.fromhex(x) for x in lst], dtype=np.float64)
        return a.reshape(shape) if shape is not None else a
    a = np.array(lst, dtype=np.int32)
    if shape is not None:
        return a.reshape(tuple(shape))
    return a.reshape(-1, 2)


def prob_dec(d, kind):
    out = {}
    for k, v in d.items():
        out[k] = dec(v["data"], kind, v["shape"])
    return out


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float64:
        return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
    return a.shape == b.shape and np.array_equal(a, b)


def fnv1a(arr):
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(arr).tobytes():
        h ^= b
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_g1_example(port):
    g = json.load(open(os.path.join(GOLD, "g1_example.json")))
    st, v, sol = port.six_solve(F64, True, [2.0, -1.0, 0.0], [[-1, 0, 0], [0, -1, 0]], None,
                                [[2, -1, 2], [1, -5, -4]])
    assert st == g["float_max"]["status"] == 0
    assert same([v], dec(g["float_max"]["v"], F64)) and v == 2.0
    assert same(sol, dec(g["float_max"]["sol"], F64))
    assert sol.tolist() == [1.5555555555555556, 1.1111111111111112, 1.0]   # example.cpp:89-93
    tg = [1, 1, 1, 1, 1, 0]
    leq = [[-1, 0, 0, 0, 0, -10], [-1, -1, 0, 0, 0, -8], [-1, -1, -1, 0, 0, -9], [-1, -1, -1, -1, 0, -11],
           [0, -1, -1, -1, -1, -13], [0, 0, -1, -1, -1, -8], [0, 0, 0, -1, -1, -5], [0, 0, 0, 0, -1, -3]]
    vc = np.zeros((5, 6), dtype=np.int32); vc[range(5), range(5)] = -1
    assert port.six_solve(RAT, True, tg, vc, None, leq)[0] == g["rat_max"]["status"] == 1
    st, v, sol = port.six_solve(RAT, False, tg, vc, None, leq)
    assert st == 0 and v.tolist() == [23, 1]                               # example.cpp:171-174
    assert same(sol, dec(g["rat_min"]["sol"], RAT))


def test_g2_two_stage_states(port):
    g = json.load(open(os.path.join(GOLD, "g2_two_stage.json")))
    n = 0
    for case in g:
        kind = case["kind"]
        if case["fam"] == "xorshift_8x16":
            leq, tgtf = gen.dense_lp_f64(8, 16)
            for s in case["states"]:
                r = port.two_stage(F64, leq, tgtf, s["K"])
                assert r["status"] == s["status"] and r["rhs"] == s["rhs"]
                assert fnv1a(r["tab"]) == s["tab_hash"]
                assert same(r["tgtf"], dec(s["tgtf"], F64))
                assert r["eq2bv"].tolist() == s["eq2bv"]
                n += 1
            continue
        p = prob_dec(case["problem"], kind)
        for s in case["states"]:
            r = port.two_stage(kind, p["leq"], p["tgtf"], s["K"])
            assert r["status"] == s["status"], (case["fam"], s["K"])
            if s["status"] == 2:
                continue
            shape = s["tab_shape"] + ([2] if kind == RAT else [])
            assert same(r["tab"], dec(s["tab"], kind, shape))
            assert same(r["tgtf"], dec(s["tgtf"], kind, None if kind == F64 else (-1, 2)))
            assert r["nvset"].tolist() == s["nvset"] and r["bvset"].tolist() == s["bvset"]
            assert r["bv2eq"].tolist() == s["bv2eq"] and r["eq2bv"].tolist() == s["eq2bv"]
            n += 1
    assert n > 60


def test_g3_six_status_objective_solution(port):
    g = json.load(open(os.path.join(GOLD, "g3_six.json")))
    statuses = set()
    for case in g:
        kind = case["kind"]
        p = prob_dec(case["problem"], kind)
        for key, is_max in (("max", True), ("min", False)):
            if key not in case:
                continue
            st, v, sol = port.six_solve(kind, is_max, p["tgtf"], p["vc"], p.get("eq"), p.get("leq"))
            w = case[key]
            assert st == w["status"]
            assert same(np.atleast_1d(v) if kind == F64 else v.reshape(-1, 2),
                        dec(w["v"], kind))
            if st == 0:
                assert same(sol, dec(w["sol"], kind) if kind == F64 else dec(w["sol"], kind))
            statuses.add(st)
    assert {0, 1, 2, 3} <= statuses      # incl. the reference's "optimal is infeasible" answers


def test_g4_rational_hashes_cross_appro(port):
    g = json.load(open(os.path.join(GOLD, "g4_rational_hash.json")))
    fired = 0
    for rec in g:
        leq, tgtf = gen.int_lp_rat(rec["m"], rec["n"])
        c0 = port.appro_count()
        r = port.two_stage(RAT, leq, tgtf, rec["K"])
        assert r["status"] == rec["status"]
        assert fnv1a(r["tab"]) == rec["tab_hash"], rec
        assert fnv1a(r["tgtf"]) == rec["tgtf_hash"]
        assert r["tgtf"][r["rhs"]].tolist() == rec["obj_const"]
        assert port.appro_count() - c0 == rec["appro_calls"]
        fired += rec["appro_calls"]
    assert fired > 10000      # the float32 rescue (rational.cpp:189-226) really is exercised


def test_g6_mip(port):
    g = json.load(open(os.path.join(GOLD, "g6_mip.json")))
    n = 0
    for case in g:
        p = prob_dec(case["problem"], RAT)
        ind = np.array(case["ind"], dtype=np.uint8) if "ind" in case else None
        for key, is_max in (("max", True), ("min", False)):
            if key not in case:
                continue
            st, v, sol = port.mip_solve(RAT, is_max, case["is_bin"], p["tgtf"], p["vc"], None, p["leq"], ind)
            w = case[key]
            assert st == w["status"]
            assert v.tolist() == w["v"]
            if st == 0:
                assert same(sol, dec(w["sol"], RAT))
            n += 1
    assert n > 40


def test_g5_lineq_and_gauss(port):
    g = json.load(open(os.path.join(GOLD, "g5_lineq.json")))
    for c in g["fme"]:
        mat = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        ok, res = port.fme(mat, c["rhs"], c["u"], c.get("dark", False))
        assert ok == c["ok"], c.get("tag")
        assert same(res, dec(c["out"], RAT, c["out_shape"])), c.get("tag")
    # the author's worked example (linsys.cpp:1035-1040): 1<=i1<=4, 5-i1<=i2<=12-i1  =>  1<=i2<=11
    c = [x for x in g["fme"] if x.get("tag") == "linsys.cpp:1035-1040"][0]
    out = dec(c["out"], RAT, c["out_shape"])
    bounds = sorted((int(r[1][0]), int(r[2][0])) for r in out if r[0][0] == 0)
    assert (-1, -1) in bounds and (1, 11) in bounds
    for c in g["reduce"]:
        mat = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        ok, res = port.reduce(mat, c["rhs"], c["inter"])
        assert ok == c["ok"] and same(res, dec(c["out"], RAT, c["out_shape"]))
    for c in g["iden"]:
        mat = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        assert same(port.remove_iden_row(mat), dec(c["out"], RAT, c["out_shape"]))
    seen = set()
    for c in g["has_solution"]:
        leq = dec(c["leq"]["data"], RAT, c["leq"]["shape"])
        nv = leq.shape[1] - 1
        vc = gen.to_rat(gen.vc_nonneg(nv, False))
        r = port.has_solution(leq, None, vc, nv, c["is_int"], c["is_unique"])
        assert r == c["result"]
        seen.add(r)
    assert seen == {0, 1}
    for c in g["calc_bound"]:
        mat = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        ok, lim = port.calc_bound(mat, c["rhs"])
        assert ok == c["ok"]
        if ok:
            for got, w in zip(lim, c["limits"]):
                want = dec(w["data"], RAT, w["shape"])
                assert (got.shape[0] == 0 and want.shape[0] == 0) or same(got, want)
    for c in g["gauss"]:
        sq = dec(c["sq"]["data"], RAT, c["sq"]["shape"])
        assert port.rat_rank(sq) == c["rank"]
        assert list(port.rat_det(sq)) == c["det"]
        ok, inv = port.rat_inv(sq)
        assert ok == c["inv_ok"]
        if ok:
            assert same(inv, dec(c["inv"], RAT, sq.shape))
        rect = dec(c["rect"]["data"], RAT, c["rect"]["shape"])
        assert port.rat_rank(rect) == c["rect_rank"]


def test_g7_intmat_rank_basis_null(port):
    g = json.load(open(os.path.join(GOLD, "g7_intmat.json")))
    for c in g["hnf"]:
        st, h, u = port.int_hnf(np.array(c["a"], dtype=np.int32))
        assert st == 0 and h.tolist() == c["h"] and u.tolist() == c["u"]
    for c in g["gcd"]:
        assert port.int_gcd(np.array(c["a"], dtype=np.int32)).tolist() == c["out"]
    ranks = set()
    for c in g["rank_basis"]:
        m = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        rk, basis = port.rat_rank_basis(m, c["unitarize"])
        want = dec(c["basis"]["data"], RAT, c["basis"]["shape"])
        assert rk == c["rank"] and ((basis.shape[0] == 0 and want.shape[0] == 0) or same(basis, want))
        ranks.add(rk < m.shape[0])
    assert ranks == {True, False}
    for c in g["null"]:
        m = dec(c["mat"]["data"], RAT, c["mat"]["shape"])
        assert same(port.rat_null(m), dec(c["ns"]["data"], RAT, c["ns"]["shape"]))
    # the reference-undefined inputs are reported, not computed (xmat.cpp:936-941, :956-980)
    assert port.int_hnf(np.array([[-1, 0, 0], [0, 1, 0]], dtype=np.int32))[0] == -7     # cols > rows, negative diagonal
    assert port.int_hnf(np.array([[1, 0], [1, 0]], dtype=np.int32))[0] == -7            # zero diagonal below row 0


def test_scalar_semantics(port):
    # Float '==' window of 1e-17 (flty.cpp:41-58)
    assert port.flt_cmp(4, 0.0, 1e-17) == 1 and port.flt_cmp(4, 0.0, 1.1e-17) == 0
    assert port.flt_cmp(4, 1e-18, -1e-18) == 0          # opposite signs never equal
    assert port.flt_cmp(1, 1e-18, 0.0) == 1             # <= is < or ==
    assert port.flt_cmp(2, 1e-18, 0.0) == 1             # > is raw
    # Rational: '==' is field-wise, '<' cross-multiplies (rational.h:80-83, rational.cpp:229-237)
    assert port.rat_cmp(4, (1, 2), (2, 4)) == 0 and port.rat_cmp(0, (1, 3), (1, 2)) == 1
    assert port.rat_op(1, (5, 5), (-3, 7)) == (-7, 3)   # x/x divided by b: unreduced reciprocal
    assert port.rat_op(0, (3, 4), (4, 3)) == (1, 1)
    assert port.rat_op(2, (1, 2), (-1, 2)) == (0, 1)
    # appro: 2^31/3 * 7/5 does not fit int32 -> float32 rescue
    n, d = port.rat_op(0, (2147483647, 3), (7, 5))
    assert d in (1, 10, 100, 1000, 10000, 100000, 1000000) or d > 0


@pytest.mark.ref
def test_live_reference_agrees_with_port(ref, port):
    """Authoring container only: a short differential run against the real reference."""
    rng = np.random.default_rng(99)
    for it in range(60):
        kind = int(rng.integers(0, 2)); fam = int(rng.integers(0, 3))
        m, nv = int(rng.integers(1, 8)), int(rng.integers(1, 8))
        p = gen.random_problem(rng, kind, fam, m, nv)
        for is_max in (True, False):
            o = port.six_solve(kind, is_max, p["tgtf"], p["vc"], p.get("eq"), p.get("leq"))
            if o[0] == -7:
                continue
            r = ref.six_solve(kind, is_max, p["tgtf"], p["vc"], p.get("eq"), p.get("leq"))
            assert r[0] == o[0] and same(r[1], o[1])
            if r[0] == 0:
                assert same(r[2], o[2])


# ---- the fixtures at the BASELINE shapes (tests/golden/g8_large.json, tools/gen_golden_large.py) -------------
def _checksum(a):
    import zlib
    a = np.ascontiguousarray(a)
    v = a.view(np.uint64).reshape(-1) if a.dtype.itemsize == 8 else a.view(np.uint32).reshape(-1).astype(np.uint64)
    return dict(crc32="%08x" % (zlib.crc32(a.tobytes()) & 0xFFFFFFFF), sum="%016x" % int(v.sum(dtype=np.uint64)),
                xor="%016x" % int(np.bitwise_xor.reduce(v)))


def test_g8_large_pins_the_oracle_at_the_baseline_shapes(port):
    """The restatement against what the real reference returned at 32 x 64 (64 LPs per family here, all 256 on
    the GPU side), at the rational 1024 x 2048 tableau (K = 8) and at the 4096 x 12289 tableau (K = 16)."""
    import zlib
    g = json.load(open(os.path.join(GOLD, "g8_large.json")))
    vc = gen.vc_nonneg(63)
    for rec in g["g3_large"]:
        leq, tg = gen.small_lp_batch_f64(256, 32, 64, rec["family"], seed=gen.XS_SEED + rec["seed_offset"])
        for b in range(0, 256, 4):
            want = rec["records"][b]
            st, v, sol = port.six_solve(F64, True, tg[b], vc, None, leq[b])
            assert st == want["status"] and float(v).hex() == want["v"], (rec["family"], b)
            if st == 0:
                assert "%08x" % (zlib.crc32(np.ascontiguousarray(sol).tobytes()) & 0xFFFFFFFF) == want["sol_crc32"]
    rec = g["g4_large"][0]
    leq, tgtf = gen.int_lp_rat(1024, 1023)
    r = port.two_stage(RAT, leq, tgtf, rec["K"])
    assert r["status"] == rec["status"] and _checksum(r["tab"]) == rec["tab"] and _checksum(r["tgtf"]) == rec["tgtf"]
    assert _checksum(r["eq2bv"].astype(np.int32)) == rec["eq2bv"]
    rec = g["g2_large"][0]
    leq, tgtf = gen.dense_lp_f64(4096, 8192)
    r = port.two_stage(F64, leq, tgtf, rec["K"])
    assert list(r["tab"].shape) == rec["tab_shape"] and r["status"] == rec["status"]
    assert _checksum(r["tab"]) == rec["tab"] and _checksum(r["tgtf"]) == rec["tgtf"]
    assert _checksum(r["eq2bv"].astype(np.int32)) == rec["eq2bv"]


def test_g9_move2var_pins_the_oracle(port):
    g = json.load(open(os.path.join(GOLD, "g9_dep.json")))
    for c in g["move2var"]:
        mat = np.array(c["mat"]["data"], dtype=np.int32).reshape(c["mat"]["shape"])
        want = np.array(c["out"]["data"], dtype=np.int32).reshape(c["out"]["shape"])
        assert np.array_equal(port.move2var(mat, c["rhs"], c["first"], c["last"]), want)


def test_g10_mip_bench_shape_pins_the_oracle(port):
    """MIP<RMat,Rational>::maxm(is_bin) at BASELINE config 5's bench shape (0-1 knapsacks of 24 variables): the
    restatement against what the real reference returned (tools/gen_golden_mip_bench.py), the inputs regenerated
    from the deterministic generator and checked by hash."""
    import hashlib
    g = json.load(open(os.path.join(GOLD, "g10_mip_bench.json")))
    leq, tgtf = gen.knapsack_batch_rat(g["nb"], g["nv"])
    assert hashlib.sha256(np.ascontiguousarray(leq).tobytes() + np.ascontiguousarray(tgtf).tobytes()).hexdigest() == g["inputs_sha256"]
    vc = gen.to_rat(gen.vc_nonneg(g["nv"], False))
    seen = set()
    for b in range(0, g["nb"], 2):
        want = g["results"][b]
        st, v, sol = port.mip_solve(RAT, True, True, tgtf[b], vc, None, leq[b])
        if want is None:
            assert st == -7, b
            continue
        assert st == want["status"] and [int(v[0]), int(v[1])] == want["v"], b
        if st == 0:
            assert [int(x) for x in np.asarray(sol).reshape(-1)] == want["sol"], b
        seen.add(st)
    assert {0, 2} <= seen


def test_end_state_and_shape_fixtures_are_complete_and_reproducible():
    """tests/golden/g12_end_states.json / g13_shapes.json (tools/gen_golden_end.py, tools/gen_golden_bench.py shape): every record
    the GPU tests and bench.py read is there, came from the real reference, and the seeded generators rebuild the inputs'
    shapes (the inputs themselves are not stored)."""
    import json
    from tools import gen
    g12 = json.load(open(os.path.join(GOLD, "g12_end_states.json")))
    for key, status in (("bench_end", 3), ("bench_six_max", 3), ("succ_two_stage", 0), ("succ_six_max", 0), ("succ_big_six_max", 0),
                        ("cover_six_min", 0), ("dense_max_256x512", 1), ("dense_min_256x512", 2), ("rational_k64", 4)):
        assert g12[key]["status"] == status, key
        assert g12[key]["reference_seconds"] > 0 and g12[key]["pivots"] > 0, key
    assert g12["bench_end"]["pivots"] == g12["bench_six_max"]["pivots"] == 4165 and g12["bench_end"]["pivots_pinned_by_reference"]
    assert g12["succ_two_stage"]["pivots_pinned_by_reference"]
    a, b = float.fromhex(g12["succ_two_stage"]["maxv"]), float.fromhex(g12["succ_six_max"]["v"])
    assert a != 0.0 and a != b and abs(a - b) <= 1e-12 * abs(a)     # (the tableau's constant and maxm's sum over sol * tgtf round differently: lpsol.h:2026-2030)
    leq, tg = gen.block_lp_f64(g12["succ_block_seeds"])
    assert list(leq.shape) == g12["succ_shape"] and leq.shape[0] >= 2048
    leq, tg = gen.cover_lp_f64(g12["cover_block_seeds"])
    assert list(leq.shape) == g12["cover_shape"] and leq.shape[0] >= 2048 and (leq[:, -1] < 0).all()
    leq, tg = gen.block_lp_f64(g12["succ_big_block_seeds"][:40], wide=True)       # (the whole LP is 366 MB: a prefix)
    assert leq.shape[1] - 1 == sum(gen.lp_block_f64(s, True)[0].shape[1] for s in g12["succ_big_block_seeds"][:40])
    assert g12["succ_big_shape"][0] >= 4096 and g12["succ_big_shape"][1] - 1 >= 8192
    g13 = json.load(open(os.path.join(GOLD, "g13_shapes.json")))
    for name, (m, n) in dict(tall=(16384, 2048), wide=(1024, 20480), square=(8192, 8192), odd_width=(4096, 4094), small=(2048, 2047)).items():
        r = g13[name]
        assert r["K"] == 256 and r["status"] == 4 and r["tab_shape"] == [m, n + m + 1], name
        assert n + m < 23171                                   # the reference's own range: (n + m)^2 * 8 B must not wrap 2^32
