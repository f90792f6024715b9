#!/usr/bin/env python3
"""Generates tests/golden/g11_bench_lp.json from the REAL reference (oracle/_ref/libxpoly_ref.so): the exact LP
bench.py times -- gen.hard_lp_f64(4096, 4095), slack tableau 4096 x 8192 fp64 -- through
xcom::SIX<FloatMat,Float>::TwoStageMethod (src/com/lpsol.h:1907-1930, loop :1039-1188) with set_param(0, K) at
K = 1024, 2048 and 3840 pivots (3840 = the end of one bench step), so the whole timed range of the headline
number is pinned to the reference, not only its first 300 pivots.
Authoring-container only (about 0.07 s per pivot: ~10 minutes). The input is not stored (seeded generator);
the fixture holds checksums of what the reference returned: CRC-32 / wrapping uint64 sum / xor of the tableau
and of the objective row, the objective constant (hex float), the basis (eq2bv) checksum and its head.
The restatement (oracle/_build) is run beside it at every K and must agree bit for bit before anything is written.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import F64, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402
from tools.gen_golden_large import checksum  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "g11_bench_lp.json")
M, N = 4096, 4095


def record(r, K, m=M, n=N):
    return dict(generator="gen.hard_lp_f64(%d, %d)" % (m, n), K=K, status=int(r["status"]), rhs=int(r["rhs"]),
                tab_shape=list(r["tab"].shape), tab=checksum(r["tab"]), tgtf=checksum(r["tgtf"]),
                obj_const=float(r["tgtf"][r["rhs"]]).hex(), eq2bv=checksum(r["eq2bv"].astype(np.int32)),
                eq2bv_head=[int(x) for x in r["eq2bv"][:32]],
                bv2eq=checksum(r["bv2eq"].astype(np.int32)),
                entered_count=int(np.sum(r["eq2bv"] < n)))


def main_cfg2b():
    """The LP of bench.py's cfg2b leg: gen.hard_lp_f64(4096, 8192), slack tableau 4096 x 12289, K = 1280 (where that leg's
    timed pass ends)."""
    ref, port = Ref(), Port()
    m, n, K = 4096, 8192, 1280
    leq, tgtf = gen.hard_lp_f64(m, n)
    t0 = time.time()
    r = ref.two_stage(F64, leq, tgtf, K)
    t1 = time.time()
    rec = record(r, K, m, n)
    del r
    p = port.two_stage(F64, leq, tgtf, K)
    t2 = time.time()
    prec = record(p, K, m, n)
    del p
    print("cfg2b K", K, "reference %.1f s, restatement %.1f s" % (t1 - t0, t2 - t1), "agree" if prec == rec else "DIFFER", flush=True)
    assert prec == rec, (prec, rec)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out["cfg2b_lp"] = [rec]
    json.dump(out, open(OUT, "w"), indent=1)
    print("written", OUT)


SHAPES = {"tall": (16384, 2048), "wide": (1024, 20480), "square": (8192, 8192), "odd_width": (4096, 4094), "small": (2048, 2047)}
SHAPES_OUT = os.path.join(ROOT, "tests", "golden", "g13_shapes.json")


def main_shape(name, K=256):
    """bench.py's `shapes` leg: gen.hard_lp_f64(m, n) at LP shapes other than the headline's -- tall,
    wide, square, a tableau of odd width, a small one -- after K = 256 pivots of the real reference's TwoStageMethod, so that the
    pivots/s reported there are of verified states. tests/golden/g13_shapes.json."""
    import fcntl
    ref, port = Ref(), Port()
    m, n = SHAPES[name]
    leq, tgtf = gen.hard_lp_f64(m, n)
    t0 = time.time()
    p = port.two_stage(F64, leq, tgtf, K)
    t1 = time.time()
    prec = record(p, K, m, n)
    del p
    print(name, (m, n), "restatement %.1f s" % (t1 - t0), flush=True)
    r = ref.two_stage(F64, leq, tgtf, K)
    t2 = time.time()
    rec = record(r, K, m, n)
    del r
    print(name, "reference %.1f s" % (t2 - t1), "agree" if prec == rec else "DIFFER", flush=True)
    assert prec == rec, (prec, rec)
    rec["reference_seconds"] = round(t2 - t1, 1)
    with open(os.path.join("/tmp", os.path.basename(SHAPES_OUT) + ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        out = json.load(open(SHAPES_OUT)) if os.path.exists(SHAPES_OUT) else {}
        out[name] = rec
        json.dump(out, open(SHAPES_OUT, "w"), indent=1)
    print("written", name, SHAPES_OUT)


def main():
    if sys.argv[1:] == ["cfg2b"]:
        return main_cfg2b()
    if len(sys.argv) > 1 and sys.argv[1] == "shape":
        return main_shape(sys.argv[2])
    ks = [int(x) for x in sys.argv[1:]] or [1024, 2048, 3840]
    ref, port = Ref(), Port()
    leq, tgtf = gen.hard_lp_f64(M, N)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    recs = {int(r["K"]): r for r in out.get("bench_lp", [])}
    for K in ks:
        t0 = time.time()
        r = ref.two_stage(F64, leq, tgtf, K)
        t1 = time.time()
        rec = record(r, K)
        del r
        p = port.two_stage(F64, leq, tgtf, K)
        t2 = time.time()
        prec = record(p, K)
        del p
        print("K", K, "reference %.1f s, restatement %.1f s" % (t1 - t0, t2 - t1), "status", rec["status"],
              "agree" if prec == rec else "DIFFER", flush=True)
        assert prec == rec, (prec, rec)
        recs[K] = rec
        out["bench_lp"] = [recs[k] for k in sorted(recs)]
        json.dump(out, open(OUT, "w"), indent=1)
    print("written", OUT)


if __name__ == "__main__":
    main()
