"""Child process of tests/test_gpu_fused_rational.py: whole Rational solves (and K-pivot prefixes) of mid-size LPs on the
device-resident loop the environment selects (XPG_R32_LOOP, XPG_R32_GENERIC_EVERY are read once per process), one JSON
line per (LP, K): status, pivots, CRC-32 of tableau / objective row / basis, and the (entering, leaving) trace."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xpoly_amd                                        # noqa: E402
from tools import gen                                   # noqa: E402

RAT = 1
# (family, rows, variables): family 0 runs its pivots in phase two (tableau compared with the oracle's); 1 and 2 spend them
# in phase one with ties and degenerate steps (status and, between the loops, the pivot trace)
CASES = [(0, 70, 200), (0, 300, 100), (0, 33, 700), (1, 130, 300), (2, 40, 520), (2, 200, 90)]
KS = (7, 12, 90)


def crc(a):
    return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def problems():
    for fam, m, nv in CASES:
        rng = np.random.default_rng(7000 + 100 * fam + m)
        yield fam, m, nv, gen.random_problem(rng, RAT, fam, m, nv, plain=True)


def small_problems():
    """The small random LPs of test_gpu_parity.py::test_two_stage_matches_oracle (whole solves: every deferred decision --
    relaxed ratio pass, disableNV, findPivotNVandBVPair -- and stage 1), which the size rule sends to the two-launch loop."""
    for fam in (0, 1, 2):
        rng = np.random.default_rng(100 + 10 * RAT + fam)
        for it in range(12):
            m, nv = int(rng.integers(1, 14)), int(rng.integers(1, 14))
            yield fam, it, gen.random_problem(rng, RAT, fam, m, nv, plain=True)


SMALL_KS = (0, 1, 2, 5, 1000)

if __name__ == "__main__":
    ctx = xpoly_amd.Context()
    six = xpoly_amd.SIX(ctx, RAT)
    if os.environ.get("XPG_FUSED_SMALL") == "1":
        for fam, it, prob in small_problems():
            for K in SMALL_KS:
                six.set_param(0, K)
                got = six.TwoStageMethod(prob["leq"], prob["tgtf"])
                rec = dict(fam=fam, it=it, K=K, status=int(got["status"]))
                if got["status"] != 2:
                    rec.update(tab=crc(got["tab"]), tgtf=crc(got["tgtf"]), eq2bv=crc(np.asarray(got["eq2bv"], dtype=np.int32)))
                print(json.dumps(rec), flush=True)
        sys.exit(0)
    for fam, m, nv, prob in problems():
        for K in KS:
            six.set_param(0, K)
            got = six.TwoStageMethod(prob["leq"], prob["tgtf"])
            rec = dict(fam=fam, m=m, nv=nv, K=K, status=int(got["status"]), trace=np.asarray(got["trace"]).reshape(-1).tolist())
            if got["status"] != 2:
                rec.update(tab=crc(got["tab"]), tgtf=crc(got["tgtf"]), eq2bv=crc(np.asarray(got["eq2bv"], dtype=np.int32)))
            print(json.dumps(rec), flush=True)
