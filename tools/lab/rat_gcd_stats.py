"""CPU study of the rational sweep's gcd loops on the cfg-4 LP (1024 x 2048, pivots 1..16): how many binary-gcd
steps each cell takes and how much of a 64-lane wave's loop time is spent on lanes already finished.
Run by hand: PYTHONPATH=. python tools/lab/rat_gcd_stats.py [K ...]"""
import sys, numpy as np
sys.path.insert(0, ".")
import bench
from tools import gen
from oracle.checker import Port, RAT

def gcd_steps(x, y):
    """gcd32 of scalar.hip.h on arrays: (gcd, steps of its do-while loop; 0 when an operand is 0)."""
    x = x.astype(np.uint64).copy(); y = y.astype(np.uint64).copy()
    steps = np.zeros(x.shape, np.int32)
    zero = (x == 0) | (y == 0)
    g0 = np.where(x == 0, y, x)
    x[zero] = 1; y[zero] = 1
    def ctz(v):
        v = v.astype(np.uint64); lsb = v & (~v + np.uint64(1))
        return np.log2(lsb.astype(np.float64)).astype(np.uint64)
    sh = ctz(x | y)
    x >>= ctz(x)
    act = np.ones(x.shape, bool)
    while act.any():
        ya = y[act]; xa = x[act]
        ya >>= ctz(ya)
        lo = np.minimum(xa, ya); hi = np.maximum(xa, ya)
        x[act] = lo; y[act] = hi - lo
        steps[act] += 1
        act2 = act.copy(); act2[act] = (hi - lo) != 0
        act = act2
    g = x << sh
    g[zero] = g0[zero]; steps[zero] = 0
    return g, steps

def main():
    Ks = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 12, 16]
    port = Port()
    leq, tgtf = gen.int_lp_rat(bench.RAT_M, bench.RAT_N)
    for K in Ks:
        a = port.two_stage(RAT, leq, tgtf, K - 1); b = port.two_stage(RAT, leq, tgtf, K)
        r = int(np.nonzero(a["eq2bv"] != b["eq2bv"])[0][0]); c = int(b["eq2bv"][r])
        T = a["tab"].astype(np.int64); e = b["tab"][r].astype(np.int64)       # scaled pivot row
        k = -T[:, c, 0], T[:, c, 1]
        m, W = T.shape[0], T.shape[1]
        rows = np.arange(m) != r
        an, ad = T[rows, :, 0], T[rows, :, 1]
        kn, kd = np.abs(k[0][rows])[:, None] + 0 * an, k[1][rows][:, None] + 0 * an
        en, ed = np.abs(e[:, 0])[None, :] + 0 * an, e[:, 1][None, :] + 0 * an
        live = (kn != 0) & (en != 0)
        g1, s1 = gcd_steps(np.where(live, kn, 0), ed); g2, s2 = gcd_steps(np.where(live, en, 0), kd)
        pm = (kn // np.maximum(g1, 1)) * (en // np.maximum(g2, 1)); pd = (kd // np.maximum(g2, 1)) * (ed // np.maximum(g1, 1))
        big = live & ((pm >= 2**31 - 1) | (pd >= 2**31 - 1))
        g3, s3 = gcd_steps(np.where(live & ~big & (an != 0), ad, 0), pd)
        tot = s1 + s2 + s3
        # waves: 64 consecutive columns of one row
        Wp = (W + 63) // 64 * 64
        def waves(s):
            z = np.zeros((s.shape[0], Wp), np.int64); z[:, :W] = s
            return z.reshape(s.shape[0], -1, 64)
        rep = []
        for name, s in (("k.num|e.den", s1), ("e.num|k.den", s2), ("a.den|p.den", s3)):
            w = waves(s); mx = w.max(axis=2)
            rep.append("%s: mean %.1f, wave max mean %.1f, lane use %.2f" % (name, s[live].mean() if live.any() else 0, mx[mx > 0].mean() if (mx > 0).any() else 0,
                                                                           w.sum() / max(1, 64 * mx.sum())))
        wl = waves(live.astype(np.int64))
        print("pivot %2d: row %d col %d; live cells %.3f of all; live lanes in live waves %.2f; appro-size products %.4f; ints(a) %.3f ints(k) %.3f ints(e) %.3f"
              % (K, r, c, live.mean(), wl.sum() / max(1, 64 * (wl.max(axis=2) > 0).sum()), big.sum() / max(1, live.sum()),
                 (ad == 1).mean(), (k[1] == 1).mean(), (e[:, 1] == 1).mean()))
        for s in rep: print("    " + s)

if __name__ == "__main__":
    main()
