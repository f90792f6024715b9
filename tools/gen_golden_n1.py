#!/usr/bin/env python3
"""Generates tests/golden/g9_dep.json from the REAL reference (oracle/_ref): Lineq::move2var on systems with
constant symbols, and DepPoly::is_empty(keepit, vc) composed from the reference's own Lineq::move2var,
Lineq::reduce and Lineq::has_solution exactly as src/eng/poly.cpp:530-573 composes them (poly.cpp itself is
outside the reference build of oracle/Makefile). Authoring-container only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle.checker import RAT, Port, Ref  # noqa: E402
from tools import gen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "g9_dep.json")


def enc(a):
    return dict(shape=list(np.asarray(a).shape), data=[int(x) for x in np.asarray(a).reshape(-1)])


def is_empty_ref(ref, port, mat, rhs_idx, vc):
    """poly.cpp:530-573 on the reference's own pieces; None where the reference is undefined."""
    cols = mat.shape[1]
    work = mat
    if rhs_idx != cols - 1:
        work = ref.move2var(mat, rhs_idx, rhs_idx + 1, cols - 1)
    ok, res = ref.reduce(work, cols - 1, True)
    if not ok:
        return 1
    if res.shape[0] == 0:
        return 0
    if rhs_idx != cols - 1:
        return -7                                   # has_solution with several constant columns: SIX::verify ASSERTs
    if vc is None:
        vc = gen.to_rat(gen.vc_nonneg(rhs_idx, False))
    if port.has_solution(res, None, vc, rhs_idx, True, True) == -7:
        return -7
    return int(not ref.has_solution(res, None, vc, rhs_idx, True, True))


def main():
    ref, port = Ref(), Port()
    rng = np.random.default_rng(20260202)
    g = dict(move2var=[], is_empty=[])
    for it in range(40):
        rows, nv, ns = int(rng.integers(1, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 4))
        m = gen.random_system(rng, rows, nv + ns)
        if it % 4 == 0:
            m[0, nv + 1] = (4, 6)                   # not in lowest terms: Matrix::mul(-1) reduces it
        out = ref.move2var(m, nv, nv + 1, nv + ns)
        g["move2var"].append(dict(mat=enc(m), rhs=nv, first=nv + 1, last=nv + ns, out=enc(out)))
    for it in range(160):
        rows, nv = int(rng.integers(2, 13)), int(rng.integers(1, 5))
        ns = int(rng.integers(0, 3)) if it % 2 else 0
        m = gen.random_system(rng, rows, nv + ns)
        m[..., 1] = 1
        vc = None
        if ns == 0 and it % 4 == 2:                 # caller-supplied variable constraints: -x_i <= c_i
            v = gen.vc_nonneg(nv, False)
            v[:, nv] = rng.integers(-1, 3, size=nv)
            vc = gen.to_rat(v)
        r = is_empty_ref(ref, port, m, nv, vc)
        g["is_empty"].append(dict(mat=enc(m), rhs=nv, vc=None if vc is None else enc(vc), empty=r))
    json.dump(g, open(OUT, "w"))
    from collections import Counter
    print("written", OUT, Counter(x["empty"] for x in g["is_empty"]))


if __name__ == "__main__":
    main()
