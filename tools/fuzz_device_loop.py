"""GPU fuzz (one-off, not part of the suite): the device-resident loop (pipelined for fp64, serial
for rational) against the CPU oracle on many small LPs of the two families that exercise its rare
branches -- dependence-test-like integer data (ties, zero pivots, pair-table exhaustion) and random
problems with phase 1 -- comparing status, tableau, objective row and basis bit for bit.

usage: python -m tools.fuzz_device_loop [n] [seed]
"""
import sys

import numpy as np

import xpoly_amd
from oracle.checker import Port
from tools import gen

F64, RAT = 0, 1
KEYS = ["tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv"]


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == np.float64:
        return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
    return a.shape == b.shape and np.array_equal(a, b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = xpoly_amd.Context(0)
    port = Port()
    hist = {}
    for it in range(n):
        kind = F64 if it % 4 else RAT
        if kind == F64 and it % 2:
            m, cols = int(rng.integers(4, 28)), int(rng.integers(4, 40))
            leqs, tgs = gen.small_lp_batch_f64(1, m, cols, family=1, seed=gen.XS_SEED + 1000 + it)
            leq, tg = leqs[0], tgs[0]
        else:
            p = gen.random_problem(rng, kind, int(rng.integers(0, 3)), int(rng.integers(1, 16)), int(rng.integers(1, 16)),
                                   plain=True)
            leq, tg = p["leq"], p["tgtf"]
        six = xpoly_amd.SIX(ctx, kind)
        for K in (int(rng.integers(1, 40)), 0xFFFFFFFF):
            want = port.two_stage(kind, leq, tg, K)
            six.set_param(0, K)
            got = six.TwoStageMethod(leq, tg)
            assert got["status"] == want["status"], (it, kind, K, got["status"], want["status"])
            hist[(kind, want["status"])] = hist.get((kind, want["status"]), 0) + 1
            if want["status"] == 2:
                continue
            for k in KEYS:
                assert same(got[k], want[k]), (it, kind, K, k)
    print("device loop fuzz ok:", n, dict(sorted(hist.items())))


if __name__ == "__main__":
    main()
