#!/usr/bin/env python3
"""Generates tests/golden/g12_end_states.json from the REAL reference (oracle/_ref/libxpoly_ref.so): full-size solves pinned at
their END -- optimum detection, the basic solution, the in-order row sums of SIX::is_feasible (src/com/lpsol.h:784-822, what
decides status 0 against 3), calcFinalSolution (:1851-1899) and the objective maxm / minm return (:1993-2033, :1662-1732) --
where every earlier fixture at these sizes stops at SIX_TIME_OUT.

    python tools/gen_golden_end.py bench_end | bench_six_max | dense_max M N | dense_min M N | succ SEEDS.json |
                                   succ_big SEEDS.json | cover SEEDS.json | rational [K]          (authoring container only)

  bench_end          TwoStageMethod on gen.hard_lp_f64(4096, 4095) -- the LP bench.py times, tableau 4096 x 8192 -- to its
                     natural end (SIX_OPTIMAL_IS_INFEASIBLE after 4165 pivots)
  bench_six_max      the same LP through SIX<FloatMat,Float>::maxm with vc = -I, no iteration limit
  dense_max/min M N  SIX::maxm / minm on gen.dense_lp_f64(M, N) (the cfg-2b recipe of SURVEY 8d), no iteration limit. NOT at
                     4096 x 8192: the recipe does not "converge in a few dozen pivots" under the reference -- rounding leaves
                     tiny positive costs, the loop goes on until the pivot-pair table is exhausted and reports SIX_UNBOUND
                     (maxm) / SIX_NO_PRI_FEASIBLE_SOL (minm): 327 766 / 505 503 pivots at 256 x 512, 1 299 430 at 512 x 1024
                     (restatement; ~4x per doubling), i.e. of the order of 1e8 pivots = months of the reference at 4096 x 8192.
                     The 256 x 512 ends are pinned instead: a third of a million pivots through every rare branch of the loop
  succ               a 2309-row fp64 LP that ends SIX_SUCC with a non-zero optimum: gen.block_lp_f64 of blocks found by a search
                     with the restatement (tools/find_blocks.py; the block seeds are part of the fixture), TwoStageMethod and maxm
  succ_big           the same construction at config 2's size (>= 4096 rows, >= 8192 variables), SIX::maxm
  cover              SIX::minm ending SIX_SUCC with a non-zero optimum on >= 2048 rows: gen.cover_lp_f64 (min c.x, A x >= b)
  rational           the cfg-4 Rational LP (gen.int_lp_rat(1024, 1023)) at K = 64 pivots

Inputs are not stored (seeded generators). Each record holds what the reference returned: status, objective bits, CRC-32 /
wrapping sum / xor of the solution and (TwoStageMethod) of the whole tableau, objective row and basis. `pivots` is the count
of SIX::pivot calls the restatement made on the same input (the reference keeps no counter); for TwoStageMethod records it is
pinned to the reference as well: with max_iter = pivots the reference must stop at SIX_TIME_OUT in the very state it ends in
(`pivots_pinned_by_reference`). The restatement (oracle/_build) runs beside the reference and must agree bit for bit before
anything is written.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import F64, RAT, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402
from tools.gen_golden_large import checksum  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "g12_end_states.json")
NO_LIMIT = 0xFFFFFFFF


def save(key, rec, **more):
    import fcntl
    with open(os.path.join("/tmp", os.path.basename(OUT) + ".lock"), "w") as lk:                 # several cases may be generated side by side
        fcntl.flock(lk, fcntl.LOCK_EX)
        out = json.load(open(OUT)) if os.path.exists(OUT) else {}
        out[key] = rec
        out.update(more)
        json.dump(out, open(OUT, "w"), indent=1)
    print("written", key, "->", OUT, flush=True)


def val(kind, v):
    return float(v).hex() if kind == F64 else [int(v[0]), int(v[1])]


def six_record(kind, st, v, sol):
    rec = dict(status=int(st), v=val(kind, v))
    if st == 0:
        rec["sol"] = checksum(np.ascontiguousarray(sol))
        nz = np.count_nonzero(sol) if kind == F64 else np.count_nonzero(sol[..., 0])
        rec["sol_nonzeros"] = int(nz)
    return rec


def ts_record(kind, r):
    rhs = int(r["rhs"])
    rec = dict(status=int(r["status"]), rhs=rhs, tab_shape=list(r["tab"].shape[:2]), tab=checksum(r["tab"]), tgtf=checksum(r["tgtf"]),
               obj_const=val(kind, r["tgtf"][rhs]), eq2bv=checksum(r["eq2bv"].astype(np.int32)),
               eq2bv_head=[int(x) for x in r["eq2bv"][:32]], bv2eq=checksum(r["bv2eq"].astype(np.int32)),
               maxv=val(kind, r["maxv"]))
    if r["status"] == 0:
        rec["sol"] = checksum(np.ascontiguousarray(r["sol"]))
    return rec


def six_case(key, generator, kind, is_max, tg, vc, leq):
    ref, port = Ref(), Port()
    t0 = time.time()
    st, v, sol = ref.six_solve(kind, is_max, tg, vc, None, leq)
    t1 = time.time()
    rec = six_record(kind, st, v, sol)
    p0 = port.pivot_count()
    pst, pv, psol = port.six_solve(kind, is_max, tg, vc, None, leq)
    t2 = time.time()
    prec = six_record(kind, pst, pv, psol)
    print(key, "reference %.1f s, restatement %.1f s" % (t1 - t0, t2 - t1), rec, "agree" if rec == prec else "DIFFER", flush=True)
    assert rec == prec, (rec, prec)
    rec.update(generator=generator, call="SIX::maxm" if is_max else "SIX::minm", max_iter="none", pivots=int(port.pivot_count() - p0),
               reference_seconds=round(t1 - t0, 1))
    save(key, rec)


def two_stage_case(key, generator, kind, leq, tg, K=NO_LIMIT, pin_pivots=True):
    ref, port = Ref(), Port()
    p0 = port.pivot_count()
    t0 = time.time()
    p = port.two_stage(kind, leq, tg, K)
    t1 = time.time()
    pivots = int(port.pivot_count() - p0)
    prec = ts_record(kind, p)
    del p
    print(key, "restatement %.1f s" % (t1 - t0), "status", prec["status"], "pivots", pivots, flush=True)
    r = ref.two_stage(kind, leq, tg, K)
    t2 = time.time()
    rec = ts_record(kind, r)
    del r
    print(key, "reference %.1f s" % (t2 - t1), "agree" if rec == prec else "DIFFER", flush=True)
    assert rec == prec, (rec, prec)
    rec.update(generator=generator, call="SIX::TwoStageMethod", max_iter="none" if K == NO_LIMIT else K, pivots=pivots,
               reference_seconds=round(t2 - t1, 1))
    if K == NO_LIMIT and pin_pivots:
        # the reference with max_iter = pivots: stops at SIX_TIME_OUT (the end is only DETECTED in iteration pivots + 1) in the very
        # state it ends in => it, too, made exactly `pivots` pivots
        r = ref.two_stage(kind, leq, tg, pivots)
        at = ts_record(kind, r)
        del r
        assert at["status"] == 4, at["status"]
        for k in ("tab", "eq2bv", "bv2eq", "obj_const"):
            assert at[k] == rec[k], k
        rec["pivots_pinned_by_reference"] = True
        print(key, "reference with max_iter = %d: SIX_TIME_OUT in the end state" % pivots, flush=True)
    save(key, rec)


def main():
    what = sys.argv[1]
    if what in ("dense_max", "dense_min"):
        m, n = int(sys.argv[2]), int(sys.argv[3])
        leq, tg = gen.dense_lp_f64(m, n)
        six_case("%s_%dx%d" % (what, m, n), "gen.dense_lp_f64(%d, %d), vc = gen.vc_nonneg(%d)" % (m, n, n), F64, what == "dense_max", tg,
                 gen.vc_nonneg(n), leq)
    elif what == "bench_end":
        leq, tg = gen.hard_lp_f64(4096, 4095)
        two_stage_case("bench_end", "gen.hard_lp_f64(4096, 4095)", F64, leq, tg)
    elif what == "bench_six_max":
        leq, tg = gen.hard_lp_f64(4096, 4095)
        six_case("bench_six_max", "gen.hard_lp_f64(4096, 4095), vc = gen.vc_nonneg(4095)", F64, True, tg, gen.vc_nonneg(4095), leq)
    elif what == "succ":
        seeds = json.load(open(sys.argv[2]))
        leq, tg = gen.block_lp_f64(seeds)
        two_stage_case("succ_two_stage", "gen.block_lp_f64(block_seeds)", F64, leq, tg)
        n = leq.shape[1] - 1
        six_case("succ_six_max", "gen.block_lp_f64(block_seeds), vc = gen.vc_nonneg(n)", F64, True, tg, gen.vc_nonneg(n), leq)
        save("succ_block_seeds", [int(s) for s in seeds], succ_shape=list(leq.shape))
    elif what == "succ_big":
        seeds = json.load(open(sys.argv[2]))
        leq, tg = gen.block_lp_f64(seeds, wide=True)
        n = leq.shape[1] - 1
        six_case("succ_big_six_max", "gen.block_lp_f64(block_seeds, wide=True), vc = gen.vc_nonneg(n)", F64, True, tg, gen.vc_nonneg(n), leq)
        save("succ_big_block_seeds", [int(s) for s in seeds], succ_big_shape=list(leq.shape))
    elif what == "cover":
        seeds = json.load(open(sys.argv[2]))
        leq, tg = gen.cover_lp_f64(seeds)
        n = leq.shape[1] - 1
        six_case("cover_six_min", "gen.cover_lp_f64(block_seeds), vc = gen.vc_nonneg(n)", F64, False, tg, gen.vc_nonneg(n), leq)
        save("cover_block_seeds", [int(s) for s in seeds], cover_shape=list(leq.shape))
    elif what == "rational":
        leq, tg = gen.int_lp_rat(1024, 1023)
        K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
        two_stage_case("rational_k%d" % K, "gen.int_lp_rat(1024, 1023)", RAT, leq, tg, K)
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
