import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.six import MIP, SIX, has_solution
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
leq = gen.to_rat(np.array([[0, 3, -3, 0, 1, -2], [3, -2, -3, -1, -2, 2], [1, 0, -1, 3, 1, -1], [-2, -3, 3, 3, -3, -1]], np.int32))
eq = gen.to_rat(np.array([[2, -2, 2, 2, 0, 2], [0, -1, 0, -2, 0, -3]], np.int32))
nv = 5
vc = gen.to_rat(gen.vc_nonneg(nv, False))
tg = gen.to_rat(np.array([1, 1, 1, 1, 1, 0], np.int32))
mip = MIP(ctx, 1); six = SIX(ctx, 1)
for is_max in (True, False):
    g = (mip.maxm if is_max else mip.minm)(tg, vc, eq, leq, False, None)
    w = port.mip_solve(1, is_max, False, tg, vc, eq, leq)
    print("mip", is_max, "gpu", g[0], np.asarray(g[1]).tolist(), "oracle", w[0], np.asarray(w[1]).tolist())
    g = (six.maxm if is_max else six.minm)(tg, vc, eq, leq)
    w = port.six_solve(1, is_max, tg, vc, eq, leq)
    print("six", is_max, "gpu", g[0], np.asarray(g[1]).tolist(), "oracle", w[0], np.asarray(w[1]).tolist())
for ii in (True, False):
    for uu in (True, False):
        print("has_solution", ii, uu, "gpu", has_solution(ctx, leq, eq, vc, nv, ii, uu), "oracle", port.has_solution(leq, eq, vc, nv, ii, uu))
