"""Debug aid: DeviceLP TwoStageMethod (Rational) on the small random LPs of tests/test_gpu_parity.py against the oracle,
printing the pivot traces where the status or the trace differs. XPG_R32_LOOP=pipe|serial selects the older loops."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import xpoly_amd
from xpoly_amd import RAT
from tools import gen
from oracle.checker import Port

ctx = xpoly_amd.Context()
port = Port()
bad = 0
for fam in (0, 1, 2):
    rng = np.random.default_rng(100 + 10 * RAT + fam)
    six = xpoly_amd.SIX(ctx, RAT)
    for it in range(12):
        m, nv = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        prob = gen.random_problem(rng, RAT, fam, m, nv, plain=True)
        for K in (0, 1, 2, 5, 1000):
            want = port.two_stage(RAT, prob["leq"], prob["tgtf"], K)
            six.set_param(0, K)
            got = six.TwoStageMethod(prob["leq"], prob["tgtf"])
            flag = "" if got["status"] == want["status"] else "  <-- BAD"
            bad += 1 if flag else 0
            print("fam", fam, "it", it, "K", K, "m", m, "nv", nv, "status", got["status"], "want", want["status"],
                  "trace", np.asarray(got["trace"]).reshape(-1).tolist(), flag)
print("bad", bad)
