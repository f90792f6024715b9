"""GPU cross-check, larger than the collected tests' (run by hand: python tests/crosscheck_mip.py [seed] [count]): xpg_mip_batch_rat32 (the tree walks on the device) against the
CPU oracle's MIP::maxm / minm -- status, value, solution -- on random integer and 0-1 problems, including 0-1
problems with more rows than columns (where the reference's equality substitution is undefined) and knapsacks of
the bench shape. XPG_MIP_DEVICE=0 checks the host controller instead."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from xpoly_amd.six import mip_batch
from oracle.checker import Port
from tools import gen

RAT = xpoly_amd.RAT
ctx = xpoly_amd.Context(0)
port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bad = 0
shapes = [(False, 3, 4), (False, 2, 6), (False, 5, 5), (False, 4, 9), (True, 1, 6), (True, 1, 10), (True, 3, 5), (True, 6, 4)]
for (is_bin, m, nv) in shapes:
    probs = [gen.random_mip(rng, m, nv, False) for _ in range(nb)]
    if is_bin:
        for p in probs:
            ub = np.zeros((nv, nv + 1), dtype=np.int32); ub[np.arange(nv), np.arange(nv)] = 1; ub[:, nv] = 1
            p["leq"] = np.concatenate([p["leq"], gen.to_rat(ub)], axis=0)
    leq = np.stack([p["leq"] for p in probs]); tg = np.stack([p["tgtf"] for p in probs])
    for is_max in (True, False):
        st, v, sol, nodes = mip_batch(ctx, is_max, is_bin, tg, leq)
        wn = 0
        for b in range(nb):
            want = port.mip_solve(RAT, is_max, is_bin, probs[b]["tgtf"], probs[b]["vc"], None, probs[b]["leq"])
            if st[b] != want[0] or not np.array_equal(v[b], want[1]) or (want[0] == 0 and not np.array_equal(sol[b], want[2])):
                bad += 1; print("mismatch", is_bin, m, nv, is_max, b, st[b], want[0])
    print("is_bin=%s %dx%d: %d problems x 2 checked (%d nodes), %d mismatches so far" % (is_bin, leq.shape[1], nv + 1, nb, nodes, bad), flush=True)
leq, tg = gen.knapsack_batch_rat(nb, 16)
st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq)
vc = gen.to_rat(gen.vc_nonneg(16, False))
for b in range(nb):
    want = port.mip_solve(RAT, True, True, tg[b], vc, None, leq[b])
    if st[b] != want[0] or not np.array_equal(v[b], want[1]) or (want[0] == 0 and not np.array_equal(sol[b], want[2])):
        bad += 1; print("mismatch knapsack", b, st[b], want[0])
print("knapsacks 16 vars: %d checked (%d nodes)" % (nb, nodes))
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
