import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.lineq import Lineq
from oracle.checker import Port
ctx = xpoly_amd.Context(0); lq = Lineq(ctx); port = Port()
m = [[[0, 1], [0, 1], [3, 1], [1, 1], [2, 1]], [[0, 1], [1, 1], [-3, 1], [2, 1], [3, 1]], [[2, 1], [0, 1], [0, 1], [1, 1], [0, 1]], [[0, 1], [-1, 1], [2, 1], [-2, 1], [8, 1]], [[2, 1], [-2, 1], [-1, 1], [2, 1], [8, 1]], [[-2, 1], [0, 1], [3, 1], [-3, 1], [0, -1]], [[0, -1], [1, 1], [-2, 1], [3, 1], [8, 1]], [[1, 1], [4, 3], [-1, 1], [3, 1], [7, 1]], [[-1, 1], [3, 1], [0, 1], [0, 1], [5, 1]], [[3, 1], [0, 1], [-2, 1], [1, 1], [-5, 1]], [[-1, 1], [1, 1], [0, 1], [-1, 1], [0, 1]], [[0, 3], [-3, 1], [3, 1], [0, 1], [5, 1]]]
a = np.array(m, np.int32)
ok, b = lq.calcBound(a[None], 4, cap_rows=2000)
wok, wb = port.calc_bound(a, 4)
print("gpu ok", ok[0], [x.shape[0] for x in b[0]], "oracle ok", wok, [x.shape[0] for x in wb])
for j in range(4):
    print(" var", j, "gpu", b[0][j].tolist(), "\n        ora", wb[j].tolist())
# the same system with the weird cell made ordinary
a2 = a.copy(); a2[5, 4] = (0, 1); a2[6, 0] = (0, 1); a2[11, 0] = (0, 1)
ok2, b2 = lq.calcBound(a2[None], 4, cap_rows=2000); wok2, wb2 = port.calc_bound(a2, 4)
print("ordinary cell: gpu ok", ok2[0], [x.shape[0] for x in b2[0]], "oracle", wok2, [x.shape[0] for x in wb2], "equal", all(np.array_equal(b2[0][j], wb2[j]) for j in range(4)))
