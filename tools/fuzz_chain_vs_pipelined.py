"""GPU fuzz (one-off, not part of the suite): the blocked loop with its persistent chain launch (lp_chain.hip.h) against the
pipelined loop, which shares none of that code, on LPs of random shapes -- among them widths that make the row stride a
multiple of 4 KiB (the chain then keeps the entering column's line in LDS) -- at random iteration limits: status, tableau,
objective row and basis bit for bit. Each loop runs in a child process of its own (the loop is chosen when a context is created
from XPG_LOOP).

usage: python -m tools.fuzz_chain_vs_pipelined [n] [seed]
"""
import os
import pickle
import subprocess
import sys

import numpy as np

KEYS = ["tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv"]


def cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        m = int(rng.integers(64, 1400))
        if it % 3 == 0:                                  # W = n + m + 1 a multiple of 512
            W = 512 * int(rng.integers(max(1, (m + 66 + 511) // 512), 6))
            nv = W - m - 1
        else:
            nv = int(rng.integers(64, 1800))
        out.append((m, nv, int(rng.integers(0, 3)), int(rng.choice([13, 97, 500, 1500, 4000])), int(rng.integers(0, 1 << 30))))
    return out


def worker(loop, n, seed, path):
    os.environ["XPG_LOOP"] = loop
    import xpoly_amd
    from tools import gen
    ctx = xpoly_amd.Context(0)
    res = []
    for (m, nv, fam, K, s) in cases(n, seed):
        if fam == 2:                                     # dependence-test-like integers: ties, closed batches, the generic pick, phase one
            leqs, tgs = gen.small_lp_batch_f64(1, m, nv + 1, family=1, seed=gen.XS_SEED + s)
            leq, tg = leqs[0], tgs[0]
        else:
            leq, tg = (gen.hard_lp_f64(m, nv) if fam == 0 else gen.dense_lp_f64(m, nv, seed=gen.XS_SEED + s))
        lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
        st = lp.two_stage(K)
        got = lp.read()
        runs = 0
        if loop == "block":
            lp.chain_aborts()
            runs = lp.chain_runs
        lp.close()
        import zlib
        res.append((st, runs, [(np.asarray(got[k]).shape, zlib.crc32(np.ascontiguousarray(got[k]).tobytes())) for k in KEYS]))
    pickle.dump(res, open(path, "wb"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    outs = {}
    for loop in ("pipe", "block"):
        path = "/tmp/fuzz_chain_%s.pkl" % loop
        subprocess.check_call([sys.executable, "-m", "tools.fuzz_chain_vs_pipelined", "--worker", loop, str(n), str(seed), path])
        outs[loop] = pickle.load(open(path, "rb"))
    bad = 0
    chain_cases = 0
    for c, a, b in zip(cases(n, seed), outs["pipe"], outs["block"]):
        chain_cases += b[1] > 0
        if a[0] != b[0] or a[2] != b[2]:
            bad += 1
            print("MISMATCH", c, a[0], b[0], [x == y for x, y in zip(a[2], b[2])])
    hist = {}
    for a in outs["pipe"]:
        hist[a[0]] = hist.get(a[0], 0) + 1
    print("chain vs pipelined fuzz: %d LPs (%d with chain launches), statuses %s, %d mismatches" % (n, chain_cases, hist, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
