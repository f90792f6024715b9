#!/usr/bin/env python3
"""Generates tests/golden/g10_mip_bench.json from the REAL reference (oracle/_ref): MIP<RMat,Rational>::maxm with
is_bin on 0-1 knapsacks of BASELINE config 5's bench shape (24 variables: 2 capacity rows + 24 bound rows), the
problems themselves coming from tools/gen.knapsack_batch_rat (xorshift64, deterministic) so that only the answers
and a hash of the inputs are stored. Problems on which the reference is undefined (the restatement says so) are
skipped, as in tools/gen_golden.py. Authoring-container only."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import RAT, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "g10_mip_bench.json")
NB, NV = 128, 24


def main():
    ref, port = Ref(), Port()
    leq, tgtf = gen.knapsack_batch_rat(NB, NV)
    vc = gen.to_rat(gen.vc_nonneg(NV, False))
    recs = []
    for b in range(NB):
        if port.mip_solve(RAT, True, True, tgtf[b], vc, None, leq[b])[0] == -7:
            recs.append(None)
            continue
        st, v, sol = ref.mip_solve(RAT, True, True, tgtf[b], vc, None, leq[b])
        r = dict(status=int(st), v=[int(v[0]), int(v[1])])
        if st == 0:
            r["sol"] = [int(x) for x in np.asarray(sol).reshape(-1)]
        recs.append(r)
    h = hashlib.sha256(np.ascontiguousarray(leq).tobytes() + np.ascontiguousarray(tgtf).tobytes()).hexdigest()
    json.dump(dict(nb=NB, nv=NV, inputs_sha256=h, results=recs), open(OUT, "w"))
    hist = {}
    for r in recs:
        k = "undefined" if r is None else r["status"]
        hist[k] = hist.get(k, 0) + 1
    print("wrote", OUT, "statuses", hist)


if __name__ == "__main__":
    main()
