#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REAL reference (oracle/_ref/libxpoly_ref.so,
built from /root/reference by `make -C oracle ref`). Authoring-container only; the
fixtures are data (inputs + the reference's outputs), committed so the oracle can be
pinned on the GPU box where the reference does not exist.

fp64 values are stored as C99 hex strings so the comparison is bit-exact.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import F64, RAT, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def enc(a, kind):
    a = np.asarray(a)
    if kind == F64:
        return [float(x).hex() for x in a.reshape(-1)]
    return [int(x) for x in a.reshape(-1)]


def fnv1a(arr):
    h = 0xcbf29ce484222325
    for b in np.ascontiguousarray(arr).tobytes():
        h ^= b
        h = (h * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def prob_enc(p, kind):
    d = {}
    for k in ("tgtf", "vc", "leq", "eq"):
        if k in p and p[k] is not None:
            a = np.asarray(p[k])
            d[k] = dict(shape=list(a.shape), data=enc(a, kind))
    return d


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Ref()
    port = Port()   # only used to skip inputs on which the reference is undefined (would crash)
    # ---- G1: the bundled example (src/example/example.cpp:54-93, :106-174) ----
    g1 = {}
    leq = np.array([[2, -1, 2], [1, -5, -4]], dtype=np.float64)
    st, v, sol = ref.six_solve(F64, True, [2.0, -1.0, 0.0], [[-1, 0, 0], [0, -1, 0]], None, leq)
    g1["float_max"] = dict(status=st, v=enc([v], F64), sol=enc(sol, F64))
    tg = [1, 1, 1, 1, 1, 0]
    leq = [[-1, 0, 0, 0, 0, -10], [-1, -1, 0, 0, 0, -8], [-1, -1, -1, 0, 0, -9], [-1, -1, -1, -1, 0, -11],
           [0, -1, -1, -1, -1, -13], [0, 0, -1, -1, -1, -8], [0, 0, 0, -1, -1, -5], [0, 0, 0, 0, -1, -3]]
    vc = np.zeros((5, 6), dtype=np.int32); vc[range(5), range(5)] = -1
    st, v, sol = ref.six_solve(RAT, True, tg, vc, None, leq)
    g1["rat_max"] = dict(status=st)
    st, v, sol = ref.six_solve(RAT, False, tg, vc, None, leq)
    g1["rat_min"] = dict(status=st, v=enc(v, RAT), sol=enc(sol, RAT))
    json.dump(g1, open(os.path.join(OUT, "g1_example.json"), "w"), indent=1)

    # ---- G2: fp64 / rational K-pivot states through SIX::TwoStageMethod ----
    rng = np.random.default_rng(20260101)
    g2 = []
    for kind in (F64, RAT):
        for fam in (0, 1, 2):
            for rep in range(3):
                m, nv = int(rng.integers(2, 9)), int(rng.integers(2, 9))
                p = gen.random_problem(rng, kind, fam, m, nv, plain=True)
                states = []
                for K in (0, 1, 2, 3, 5, 8, 1000):
                    r = ref.two_stage(kind, p["leq"], p["tgtf"], K)
                    s = dict(K=K, status=r["status"], rhs=r["rhs"])
                    if r["status"] != 2:
                        s.update(tab=enc(r["tab"], kind), tab_shape=list(r["tab"].shape[:2]),
                                 tgtf=enc(r["tgtf"], kind), nvset=r["nvset"].tolist(),
                                 bvset=r["bvset"].tolist(), bv2eq=r["bv2eq"].tolist(),
                                 eq2bv=r["eq2bv"].tolist())
                        if r["status"] == 0:
                            s.update(maxv=enc([r["maxv"]] if kind == F64 else r["maxv"], kind))
                    states.append(s)
                g2.append(dict(kind=kind, fam=fam, problem=prob_enc(p, kind), states=states))
    # the 8x16 xorshift LP quoted in SURVEY.md section 8c (checksum states)
    leq, tgtf = gen.dense_lp_f64(8, 16)
    states = []
    for K in (0, 1, 2, 3, 4, 5, 6, 1000):
        r = ref.two_stage(F64, leq, tgtf, K)
        states.append(dict(K=K, status=r["status"], rhs=r["rhs"], tab_hash=fnv1a(r["tab"]),
                           tgtf=enc(r["tgtf"], F64), eq2bv=r["eq2bv"].tolist()))
    g2.append(dict(kind=F64, fam="xorshift_8x16", states=states))
    json.dump(g2, open(os.path.join(OUT, "g2_two_stage.json"), "w"))

    # ---- G3: status / objective / solution of small LPs, maxm and minm ----
    g3 = []
    for kind in (F64, RAT):
        for fam in (0, 1, 2, 3):
            for rep in range(16):
                m, nv = int(rng.integers(1, 8)), int(rng.integers(1, 8))
                p = gen.random_problem(rng, kind, fam, m, nv)
                rec = dict(kind=kind, fam=fam, problem=prob_enc(p, kind))
                for is_max in (True, False):
                    vcm = np.asarray(p["vc"])[..., 0] if kind == RAT else np.asarray(p["vc"])
                    has_free = bool(np.any(np.all(vcm[:, :-1] == 0, axis=0)))
                    if has_free:
                        continue   # x86-64 reference undefined: vcmap filled by sete()
                    if port.six_solve(kind, is_max, p["tgtf"], p["vc"], p.get("eq"), p.get("leq"))[0] == -7:
                        continue
                    st, v, sol = ref.six_solve(kind, is_max, p["tgtf"], p["vc"], p.get("eq"), p.get("leq"))
                    r = dict(status=st, v=enc([v] if kind == F64 else v, kind))
                    if st == 0:
                        r["sol"] = enc(sol, kind)
                    rec["max" if is_max else "min"] = r
                if "max" in rec:
                    g3.append(rec)
    json.dump(g3, open(os.path.join(OUT, "g3_six.json"), "w"))

    # ---- G4: rational K-pivot hashes spanning the first 'appro' activations ----
    g4 = []
    for (m, n) in ((12, 24), (24, 48)):
        leq, tgtf = gen.int_lp_rat(m, n)
        c0 = ref.appro_count()
        for K in (1, 2, 4, 8, 12, 16, 24):
            r = ref.two_stage(RAT, leq, tgtf, K)
            g4.append(dict(m=m, n=n, K=K, status=r["status"], tab_hash=fnv1a(r["tab"]),
                           tgtf_hash=fnv1a(r["tgtf"]), obj_const=enc(r["tgtf"][r["rhs"]], RAT),
                           appro_calls=int(ref.appro_count() - c0)))
            c0 = ref.appro_count()
    json.dump(g4, open(os.path.join(OUT, "g4_rational_hash.json"), "w"), indent=1)

    # ---- G6: MIP<RMat,Rational> on small integer and 0-1 instances ----
    g6 = []
    for rep in range(40):
        m, nv = int(rng.integers(1, 6)), int(rng.integers(1, 7))
        is_bin = bool(rng.integers(0, 2))
        p = gen.random_mip(rng, m, nv, is_bin)
        rec = dict(is_bin=is_bin, problem=prob_enc(p, RAT))
        if "ind" in p:
            rec["ind"] = p["ind"].tolist()
        for is_max in (True, False):
            if port.mip_solve(RAT, is_max, is_bin, p["tgtf"], p["vc"], None, p["leq"], p.get("ind"))[0] == -7:
                continue
            st, v, sol = ref.mip_solve(RAT, is_max, is_bin, p["tgtf"], p["vc"], None, p["leq"], p.get("ind"))
            r = dict(status=st, v=enc(v, RAT))
            if st == 0:
                r["sol"] = enc(sol, RAT)
            rec["max" if is_max else "min"] = r
        g6.append(rec)
    json.dump(g6, open(os.path.join(OUT, "g6_mip.json"), "w"))

    # ---- G5: Lineq::reduce / fme / removeIdenRow / has_solution, Matrix<Rational> rank/det/inv ----
    g5 = dict(fme=[], reduce=[], iden=[], has_solution=[], gauss=[])
    # the worked FME example of src/com/linsys.cpp:645-655: eliminate x from
    #   -3x-4y<=-16, 4x-7y<=20, 4x+7y<=56, -2x+3y<=9
    ex = gen.to_rat(np.array([[-3, -4, -16], [4, -7, 20], [4, 7, 56], [-2, 3, 9]], dtype=np.int32))
    ok, res = ref.fme(ex, 2, 0)
    g5["fme"].append(dict(tag="linsys.cpp:645-655", mat=dict(shape=list(ex.shape), data=enc(ex, RAT)), rhs=2, u=0,
                          ok=ok, out_shape=list(res.shape), out=enc(res, RAT)))
    # calcBound's example (linsys.cpp:1035-1040): 1<=i1<=4, 5-i1<=i2<=12-i1; eliminate i1 -> 1<=i2<=11
    ex = gen.to_rat(np.array([[-1, 0, -1], [1, 0, 4], [-1, -1, -5], [1, 1, 12]], dtype=np.int32))
    ok, res = ref.fme(ex, 2, 0)
    g5["fme"].append(dict(tag="linsys.cpp:1035-1040", mat=dict(shape=list(ex.shape), data=enc(ex, RAT)), rhs=2, u=0,
                          ok=ok, out_shape=list(res.shape), out=enc(res, RAT)))
    for rep in range(48):
        rows, nv = int(rng.integers(1, 12)), int(rng.integers(1, 7))
        mat = gen.random_system(rng, rows, nv)
        u = int(rng.integers(0, nv)); dark = bool(rng.integers(0, 2))
        ok, res = ref.fme(mat, nv, u, dark)
        g5["fme"].append(dict(mat=dict(shape=list(mat.shape), data=enc(mat, RAT)), rhs=nv, u=u, dark=dark,
                              ok=ok, out_shape=list(res.shape), out=enc(res, RAT)))
        for inter in (True, False):
            ok, res = ref.reduce(mat, nv, inter)
            g5["reduce"].append(dict(mat=dict(shape=list(mat.shape), data=enc(mat, RAT)), rhs=nv, inter=inter,
                                     ok=ok, out_shape=list(res.shape), out=enc(res, RAT)))
        res = ref.remove_iden_row(mat)
        g5["iden"].append(dict(mat=dict(shape=list(mat.shape), data=enc(mat, RAT)), out_shape=list(res.shape), out=enc(res, RAT)))
        sysm, vc = gen.random_feas(rng, int(rng.integers(1, 7)), int(rng.integers(1, 5)))
        for ii in (True, False):
            for uu in (True, False):
                if port.has_solution(sysm, None, vc, sysm.shape[1] - 1, ii, uu) == -7:
                    continue
                r = ref.has_solution(sysm, None, vc, sysm.shape[1] - 1, ii, uu)
                g5["has_solution"].append(dict(leq=dict(shape=list(sysm.shape), data=enc(sysm, RAT)),
                                               is_int=ii, is_unique=uu, result=r))
        sq = gen.random_square(rng, int(rng.integers(1, 7)))
        okinv, inv = ref.rat_inv(sq)
        rk = gen.random_system(rng, int(rng.integers(1, 7)), int(rng.integers(1, 7)))
        g5["gauss"].append(dict(sq=dict(shape=list(sq.shape), data=enc(sq, RAT)), rank=ref.rat_rank(sq),
                                det=list(ref.rat_det(sq)), inv_ok=okinv, inv=enc(inv, RAT) if okinv else None,
                                rect=dict(shape=list(rk.shape), data=enc(rk, RAT)), rect_rank=ref.rat_rank(rk)))
    # Lineq::calcBound incl. the authors' example (linsys.cpp:1035-1040)
    g5["calc_bound"] = []
    ex = gen.to_rat(np.array([[-1, 0, -1], [1, 0, 4], [-1, -1, -5], [1, 1, 12]], dtype=np.int32))
    systems = [ex] + [gen.random_system(rng, int(rng.integers(2, 7)), int(rng.integers(1, 5))) for _ in range(24)]
    for mat in systems:
        nv = mat.shape[1] - 1
        ok, lim = ref.calc_bound(mat, nv)
        g5["calc_bound"].append(dict(mat=dict(shape=list(mat.shape), data=enc(mat, RAT)), rhs=nv, ok=ok,
                                     limits=[dict(shape=list(l.shape), data=enc(l, RAT)) for l in lim] if ok else None))
    json.dump(g5, open(os.path.join(OUT, "g5_lineq.json"), "w"))

    # G7: INTMat::hnf / gcd (xmat.cpp:912-1030), rank with basis, null (matt.h:2546-2726). Only inputs
    # on which the reference is defined (the port says which) are run through it.
    rng = np.random.default_rng(77)
    g7 = dict(hnf=[], gcd=[], rank_basis=[], null=[])
    while len(g7["hnf"]) < 40:
        rows, cols = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        a = rng.integers(-6, 7, size=(rows, cols)).astype(np.int32)
        if port.int_hnf(a)[0] != 0:
            continue
        _, h, u = ref.int_hnf(a)
        g7["hnf"].append(dict(a=a.tolist(), h=h.tolist(), u=u.tolist()))
    for _ in range(30):
        rows, cols = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        a = (rng.integers(-5, 6, size=(rows, cols)) * rng.integers(1, 7, size=(rows, 1))).astype(np.int32)
        g7["gcd"].append(dict(a=a.tolist(), out=ref.int_gcd(a).tolist()))
    for it in range(30):
        rows, cols = int(rng.integers(1, 6)), int(rng.integers(2, 7))
        m = gen.random_system(rng, rows, cols - 1)
        if it % 3 == 0 and rows > 1:
            m[-1] = m[0]
        for unit in (True, False):
            rk, b = ref.rat_rank_basis(m, unit)
            g7["rank_basis"].append(dict(mat=dict(shape=list(m.shape), data=enc(m, RAT)), unitarize=unit, rank=rk,
                                         basis=dict(shape=list(b.shape), data=enc(b, RAT))))
        ns = ref.rat_null(m)
        g7["null"].append(dict(mat=dict(shape=list(m.shape), data=enc(m, RAT)), ns=dict(shape=list(ns.shape), data=enc(ns, RAT))))
    json.dump(g7, open(os.path.join(OUT, "g7_intmat.json"), "w"))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
