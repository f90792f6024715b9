"""Child process of tests/test_gpu_batch_slices.py: batches of small LPs through xpg_six_batch_* with whatever
XPG_BATCH_SLICE / XPG_BATCH_SLICE_FORCE the environment sets (read once per process); one JSON line per case with
SHA-256 of the status / value / solution arrays, plus the first LPs' results in full for the oracle comparison."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xpoly_amd                                        # noqa: E402
from tools import gen                                   # noqa: E402

F64, RAT = 0, 1
# (kind, family, LPs, is_max, iteration limit): the limits of the last two cases end some LPs in SIX_TIME_OUT, in stage 1's
# solve (status 2) or in their own (status 4), in the middle of a slice
CASES = [(F64, 0, 3072, 1, 0xFFFFFFFF), (F64, 1, 3072, 1, 0xFFFFFFFF), (F64, 1, 2048, 0, 0xFFFFFFFF), (RAT, 1, 1536, 1, 0xFFFFFFFF),
         (F64, 0, 2048, 1, 301), (F64, 1, 2048, 1, 1100)]
HEAD = 10
SMALL = os.environ.get("XPG_SLICE_TEST_SMALL") == "1"      # the forced tiny slices: 384 LPs per case


def problems():
    for kind, fam, nb, is_max, limit in CASES:
        leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam, seed=gen.XS_SEED + 4242 + 7 * fam + nb + (limit & 1023))
        if SMALL:
            nb, leq, tg = 384, leq[:384], tg[:384]
        if kind == RAT:
            leq, tg = gen.to_rat(leq.astype(np.int32)), gen.to_rat(tg.astype(np.int32))
        yield kind, fam, nb, is_max, limit, leq, tg


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


if __name__ == "__main__":
    ctx = xpoly_amd.Context()
    for kind, fam, nb, is_max, limit, leq, tg in problems():
        st, v, sol = ctx.six_batch(kind, is_max, tg, leq, max_iter=limit)
        ok = st == 0
        solm = np.where(ok.reshape((-1,) + (1,) * (sol.ndim - 1)), sol, 0)      # (sol is left alone where the status is not 0)
        print(json.dumps(dict(kind=kind, fam=fam, nb=nb, is_max=is_max, limit=limit, status=sha(st), v=sha(v), sol=sha(solm),
                              hist=np.bincount(st, minlength=5).tolist(),
                              head_status=st[:HEAD].tolist(), head_v=np.asarray(v[:HEAD]).tolist())), flush=True)
