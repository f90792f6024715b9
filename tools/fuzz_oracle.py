#!/usr/bin/env python3
"""Differential fuzz: CPU restatement (oracle/oracle.cpp) vs the real reference
(oracle/_ref/libxpoly_ref.so). Authoring-container tool; run after `make -C
oracle ref port`. Exits non-zero at the first mismatch and prints the case.

    python tools/fuzz_oracle.py [scalars|six|stage|mip|lineq|all] [--n N] [--seed S]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.checker import F64, RAT, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float64:
        return np.array_equal(a.view(np.uint64), b.view(np.uint64))
    return np.array_equal(a, b)


def fuzz_scalars(ref, port, n, rng):
    pools = [
        lambda: int(rng.integers(-20, 21)),
        lambda: int(rng.integers(-100000, 100001)),
        lambda: int(rng.integers(-2**31 + 1, 2**31 - 1)),
        lambda: int(rng.choice([0, 1, -1, 2**31 - 1, -(2**31 - 1), 2**29, 2**29 - 1, 1000000])),
    ]
    for it in range(n):
        p = pools[int(rng.integers(0, len(pools)))]
        q = pools[int(rng.integers(0, len(pools)))]
        a = (p(), q()); b = (q(), p())
        if it % 3 == 0:
            a = (a[0], abs(a[1]) or 1); b = (b[0], abs(b[1]) or 1)
        for op in range(5):
            if op == 1 and b[0] == 0 and a[0] != a[1] and a[0] != 0:
                pass  # division by 0/x still goes through integer math only
            r, o = ref.rat_op(op, a, b), port.rat_op(op, a, b)
            assert r == o, ("rat_op", op, a, b, r, o)
        for c in range(6):
            r, o = ref.rat_cmp(c, a, b), port.rat_cmp(c, a, b)
            assert r == o, ("rat_cmp", c, a, b, r, o)
    vals = [0.0, -0.0, 1e-17, 1.1e-17, -1e-17, 9e-18, 1.0, 1.0 + 2.3e-16, -1.0, 5e-324, 1e300, -3.5]
    for x in vals:
        for y in vals:
            for c in range(6):
                assert ref.flt_cmp(c, x, y) == port.flt_cmp(c, x, y), ("flt_cmp", c, x, y)
    print("scalars ok:", n)


def check_six(ref, port, kind, is_max, prob, tag, max_iter=0xFFFFFFFF):
    o = port.six_solve(kind, is_max, prob["tgtf"], prob["vc"], prob.get("eq"), prob.get("leq"), max_iter)
    if o[0] == -7:
        return "undef"
    r = ref.six_solve(kind, is_max, prob["tgtf"], prob["vc"], prob.get("eq"), prob.get("leq"), max_iter)
    ok = r[0] == o[0] and same(r[1], o[1]) and (r[0] != 0 or same(r[2], o[2]))
    if not ok:
        print("MISMATCH", tag, "kind", kind, "is_max", is_max)
        print(" ref :", r)
        print(" port:", o)
        np.save("/tmp/fuzz_fail.npy", prob, allow_pickle=True)
        raise SystemExit(1)
    return r[0]


def fuzz_six(ref, port, n, rng):
    hist = {}
    for it in range(n):
        kind = int(rng.integers(0, 2))
        fam = int(rng.integers(0, 4))
        m = int(rng.integers(1, 9)); nv = int(rng.integers(1, 9))
        prob = gen.random_problem(rng, kind, fam, m, nv)
        for is_max in (True, False):
            st = check_six(ref, port, kind, is_max, prob, (it, fam, m, nv))
            hist[(kind, is_max, st)] = hist.get((kind, is_max, st), 0) + 1
    print("six ok:", n, dict(sorted(hist.items(), key=str)))


def fuzz_stage(ref, port, n, rng):
    keys = ["tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv", "maxv"]
    for it in range(n):
        kind = int(rng.integers(0, 2))
        fam = int(rng.integers(0, 3))
        m = int(rng.integers(1, 12)); nv = int(rng.integers(1, 12))
        prob = gen.random_problem(rng, kind, fam, m, nv, plain=True)
        for K in (0, 1, 2, 3, 5, 8, 1000):
            r = ref.two_stage(kind, prob["leq"], prob["tgtf"], K)
            o = port.two_stage(kind, prob["leq"], prob["tgtf"], K)
            ok = r["status"] == o["status"] and r["rhs"] == o["rhs"]
            if ok and r["status"] != 2:
                ok = all(same(r[k], o[k]) for k in keys)
            if not ok:
                print("MISMATCH two_stage", it, kind, fam, m, nv, "K", K, r["status"], o["status"])
                for k in keys:
                    if not same(r[k], o[k]):
                        print(k, "\n ref", r[k], "\n port", o[k])
                np.save("/tmp/fuzz_fail.npy", prob, allow_pickle=True)
                raise SystemExit(1)
    print("two_stage ok:", n)


def fuzz_mip(ref, port, n, rng):
    hist = {}
    for it in range(n):
        m = int(rng.integers(1, 6)); nv = int(rng.integers(1, 7))
        is_bin = bool(rng.integers(0, 2))
        prob = gen.random_mip(rng, m, nv, is_bin)
        for is_max in (True, False):
            o = port.mip_solve(RAT, is_max, is_bin, prob["tgtf"], prob["vc"], prob.get("eq"), prob["leq"], prob.get("ind"))
            if o[0] == -7:
                hist["undef"] = hist.get("undef", 0) + 1
                continue
            r = ref.mip_solve(RAT, is_max, is_bin, prob["tgtf"], prob["vc"], prob.get("eq"), prob["leq"], prob.get("ind"))
            ok = r[0] == o[0] and same(r[1], o[1]) and (r[0] != 0 or same(r[2], o[2]))
            if not ok:
                print("MISMATCH mip", it, m, nv, is_bin, is_max, "\n ref", r, "\n port", o)
                np.save("/tmp/fuzz_fail.npy", prob, allow_pickle=True)
                raise SystemExit(1)
            hist[(is_bin, is_max, r[0])] = hist.get((is_bin, is_max, r[0]), 0) + 1
    print("mip ok:", n, dict(sorted(hist.items(), key=str)))


def fuzz_lineq(ref, port, n, rng):
    for it in range(n):
        rows = int(rng.integers(1, 10)); nv = int(rng.integers(1, 6))
        mat = gen.random_system(rng, rows, nv)
        rhs = nv
        r = ref.remove_iden_row(mat); o = port.remove_iden_row(mat)
        assert same(r, o), ("remove_iden_row", it, mat[..., 0], r, o)
        for inter in (True, False):
            r = ref.reduce(mat, rhs, inter); o = port.reduce(mat, rhs, inter)
            assert r[0] == o[0] and same(r[1], o[1]), ("reduce", it, inter, mat[..., 0], r, o)
        u = int(rng.integers(0, nv))
        r = ref.fme(mat, rhs, u); o = port.fme(mat, rhs, u)
        assert r[0] == o[0] and same(r[1], o[1]), ("fme", it, u, mat[..., 0], r, o)
        if rows <= 6 and nv <= 4:
            r = ref.calc_bound(mat, rhs); o = port.calc_bound(mat, rhs)
            assert r[0] == o[0], ("calc_bound", it, mat[..., 0], r[0], o[0])
            if r[0]:
                assert all(same(a, b) or (a.shape[0] == 0 and b.shape[0] == 0) for a, b in zip(r[1], o[1])), \
                    ("calc_bound", it, mat[..., 0], r[1], o[1])
        sq = gen.random_square(rng, int(rng.integers(1, 6)))
        assert ref.rat_rank(sq) == port.rat_rank(sq), ("rank", sq[..., 0])
        assert ref.rat_det(sq) == port.rat_det(sq), ("det", sq[..., 0], ref.rat_det(sq), port.rat_det(sq))
        ri, oi = ref.rat_inv(sq), port.rat_inv(sq)
        assert ri[0] == oi[0] and (not ri[0] or same(ri[1], oi[1])), ("inv", sq[..., 0], ri, oi)
        rk = gen.random_system(rng, int(rng.integers(1, 7)), int(rng.integers(1, 7)))
        assert ref.rat_rank(rk) == port.rat_rank(rk), ("rank", rk[..., 0])
        sysm, vc = gen.random_feas(rng, int(rng.integers(1, 7)), int(rng.integers(1, 5)))
        for ii in (True, False):
            for uu in (True, False):
                o = port.has_solution(sysm, None, vc, sysm.shape[1] - 1, ii, uu)
                if o == -7:
                    continue
                r = ref.has_solution(sysm, None, vc, sysm.shape[1] - 1, ii, uu)
                assert r == o, ("has_solution", it, ii, uu, sysm[..., 0], r, o)
    print("lineq ok:", n)


def fuzz_intmat(ref, port, n, rng):
    """N3: rank-with-basis, null, INTMat::hnf / gcd. The port is asked first; where it says the
    reference is undefined (SIGFPE / out-of-bounds read) the reference is not run."""
    undefined = 0
    for it in range(n):
        rows, cols = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        rk = gen.random_system(rng, rows, cols - 1) if cols > 1 else gen.random_square(rng, 1)
        if it % 3 == 0 and rk.shape[0] > 1:           # dependent rows: rank < rows
            rk[-1] = rk[0]
        for unit in (True, False):
            r = ref.rat_rank_basis(rk, unit); o = port.rat_rank_basis(rk, unit)
            assert r[0] == o[0] and (same(r[1], o[1]) or (r[1].shape[0] == 0 and o[1].shape[0] == 0)), \
                ("rank_basis", it, unit, rk[..., 0], r, o)
        assert same(ref.rat_null(rk), port.rat_null(rk)), ("null", it, rk[..., 0])
        lo = int(rng.choice([2, 4, 10]))
        a = rng.integers(-lo, lo + 1, size=(rows, cols)).astype(np.int32)
        if it % 4 == 0:
            a[:, int(rng.integers(0, cols))] = 0
        if it % 5 == 0 and rows > 1:
            a[int(rng.integers(0, rows))] = 0
        k = int(rng.integers(1, 6))
        assert same(ref.int_gcd(a * k), port.int_gcd(a * k)), ("gcd", it, a * k)
        st, h, u = port.int_hnf(a)
        if st == -7:
            undefined += 1
            continue
        rs, rh, ru = ref.int_hnf(a)
        assert same(rh, h) and same(ru, u), ("hnf", it, a, rh, h, ru, u)
        prod = (a.astype(np.int64) @ u.astype(np.int64)).astype(np.int32)      # the INT ring is Z/2^32
        assert np.array_equal(prod, h), ("h = a*u", it, a, h, u)
    print("intmat ok:", n, "reference-undefined hnf inputs skipped:", undefined)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ref, port = Ref(), Port()
    todo = ["scalars", "six", "stage", "mip", "lineq", "intmat"] if a.what == "all" else [a.what]
    for w in todo:
        {"scalars": fuzz_scalars, "six": fuzz_six, "stage": fuzz_stage, "mip": fuzz_mip,
         "lineq": fuzz_lineq, "intmat": fuzz_intmat}[w](ref, port, a.n, rng)
