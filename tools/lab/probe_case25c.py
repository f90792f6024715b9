import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from xpoly_amd.six import SIX
from tools import gen
ctx = xpoly_amd.Context(0)
La = np.load(os.path.join(root, "tools", "lab", "_data", "case25_folded.npy"))
vc = gen.to_rat(gen.vc_nonneg(5, False)); tg = gen.to_rat(np.array([1, 1, 1, 1, 1, 0], np.int32))
six = SIX(ctx, 1)
g = six.maxm(tg, vc, None, La); ctx.sync()
print("six maxm", g[0], np.asarray(g[1]).tolist())
