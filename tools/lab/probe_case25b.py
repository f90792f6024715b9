import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.six import SIX
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
La = np.load(os.path.join(root, "tools", "lab", "_data", "case25_folded.npy"))
nv = 5
vc = gen.to_rat(gen.vc_nonneg(nv, False))
tg = gen.to_rat(np.array([1, 1, 1, 1, 1, 0], np.int32))
six = SIX(ctx, 1)
for is_max in (True, False):
    g = (six.maxm if is_max else six.minm)(tg, vc, None, La)
    w = port.six_solve(1, is_max, tg, vc, None, La)
    print("folded six", is_max, "gpu", g[0], np.asarray(g[1]).tolist(), "oracle", w[0], np.asarray(w[1]).tolist())
for K in (0, 1, 2, 3, 5, 8):
    six.set_param(0, K)
    g = six.TwoStageMethod(La, tg); w = port.two_stage(1, La, tg, K)
    same = all(np.array_equal(g[k], w[k]) for k in ("tab", "tgtf", "eq2bv")) if g["status"] == w["status"] and w["status"] != 2 else None
    print("two_stage K", K, "gpu", g["status"], "oracle", w["status"], "state equal", same)
print("status by iteration limit (LDS kernel vs oracle):")
for K in list(range(0, 14)) + [20, 50, 1000]:
    six.set_param(0, K)
    g = six.maxm(tg, vc, None, La)
    w = port.six_solve(1, True, tg, vc, None, La, max_iter=K)
    print("  K %4d: gpu %d %s oracle %d %s" % (K, g[0], np.asarray(g[1]).tolist(), w[0], np.asarray(w[1]).tolist()))
