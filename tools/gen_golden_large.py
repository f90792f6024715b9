#!/usr/bin/env python3
"""Generates tests/golden/g8_large.json from the REAL reference (oracle/_ref/libxpoly_ref.so) at the
BASELINE shapes (SURVEY 8c G2 / G3 / G4 as specified). Authoring-container only.
Inputs are not stored: they are the seeded generators of tools/gen.py (named in each record); the fixture
holds what the reference returned -- statuses, objectives (hex floats), bases, and checksums of the big
arrays (CRC-32 of the raw bytes, wrapping uint64 sum, xor).

  G3L  256 LPs per family at 32 x 64 (dense positive, dependence-test-like), SIX<FloatMat,Float>::maxm
  G4L  exact rational simplex, tableau 1024 x 2048 (m = 1024, n = 1023), K = 8 and 16 pivots
  G2L  cfg 2b: LP m = 4096, n = 8192 (slack tableau 4096 x 12289), K = 16, 32, 48 pivots
"""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import F64, RAT, Ref  # noqa: E402
from tools import gen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "g8_large.json")


def checksum(a):
    a = np.ascontiguousarray(a)
    v = a.view(np.uint64).reshape(-1) if a.dtype.itemsize == 8 else a.view(np.uint32).reshape(-1).astype(np.uint64)
    return dict(crc32="%08x" % (zlib.crc32(a.tobytes()) & 0xFFFFFFFF), sum="%016x" % int(v.sum(dtype=np.uint64)),
                xor="%016x" % int(np.bitwise_xor.reduce(v)))


def main():
    which = set(sys.argv[1:]) or {"g3", "g4", "g2"}
    ref = Ref()
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if "g3" in which:
        g3 = []
        vc = gen.vc_nonneg(63)
        for fam in (0, 1):
            seed = gen.XS_SEED + 4242 + fam
            leq, tg = gen.small_lp_batch_f64(256, 32, 64, fam, seed=seed)
            t0 = time.time()
            recs = []
            for b in range(256):
                st, v, sol = ref.six_solve(F64, True, tg[b], vc, None, leq[b])
                r = dict(status=int(st), v=float(v).hex())
                if st == 0:
                    r["sol_crc32"] = "%08x" % (zlib.crc32(np.ascontiguousarray(sol).tobytes()) & 0xFFFFFFFF)
                recs.append(r)
            print("g3 family", fam, "%.1f s" % (time.time() - t0), flush=True)
            g3.append(dict(generator="gen.small_lp_batch_f64(256, 32, 64, %d, seed=gen.XS_SEED + %d)" % (fam, 4242 + fam),
                           family=fam, seed_offset=4242 + fam, records=recs))
        out["g3_large"] = g3
        json.dump(out, open(OUT, "w"))
    if "g4" in which:
        g4 = []
        leq, tgtf = gen.int_lp_rat(1024, 1023)
        for K in (8, 16):
            c0 = ref.appro_count()
            t0 = time.time()
            r = ref.two_stage(RAT, leq, tgtf, K)
            print("g4 K", K, "%.1f s" % (time.time() - t0), flush=True)
            g4.append(dict(generator="gen.int_lp_rat(1024, 1023)", K=K, status=int(r["status"]), rhs=int(r["rhs"]),
                           tab=checksum(r["tab"]), tgtf=checksum(r["tgtf"]),
                           obj_const=[int(x) for x in r["tgtf"][r["rhs"]]], eq2bv=checksum(r["eq2bv"].astype(np.int32)),
                           eq2bv_head=r["eq2bv"][:32].tolist(), appro_calls=int(ref.appro_count() - c0)))
        out["g4_large"] = g4
        json.dump(out, open(OUT, "w"))
    if "g2" in which:
        g2 = []
        leq, tgtf = gen.dense_lp_f64(4096, 8192)
        for K in (16, 32, 48):
            t0 = time.time()
            r = ref.two_stage(F64, leq, tgtf, K)
            print("g2 K", K, "%.1f s" % (time.time() - t0), "status", r["status"], flush=True)
            g2.append(dict(generator="gen.dense_lp_f64(4096, 8192)", K=K, status=int(r["status"]), rhs=int(r["rhs"]),
                           tab_shape=list(r["tab"].shape), tab=checksum(r["tab"]), tgtf=checksum(r["tgtf"]),
                           obj_const=float(r["tgtf"][r["rhs"]]).hex(), eq2bv=checksum(r["eq2bv"].astype(np.int32)),
                           entered=sorted(int(x) for x in r["eq2bv"] if x < 8192)))
            del r
        out["g2_large"] = g2
        json.dump(out, open(OUT, "w"))
    print("written", OUT)


if __name__ == "__main__":
    main()
