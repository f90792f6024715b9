import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import xpoly_amd
from tools import gen
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
inf = np.inf
leq = np.array([[1.0, inf, 0.0, -2.0, 0.0, -2.0, -2.0], [-2.0, 0.0, 5.0, 1.0, 0.0, 0.0, 2.0], [5.0, 0.0, 0.0, -2.0, 0.0, -3.0, 4.0]]); tg = np.array([1.0, 4.0, 2.0, 2.0, 4.0, 0.0, 0.0])
six = xpoly_amd.SIX(ctx, 0)
if os.environ.get("XPG_FORCE_DEVICE_LP"):
    for K in (1, 2, 3, 1000):
        six.set_param(0, K); g = six.TwoStageMethod(leq, tg); ctx.sync(); w = port.two_stage(0, leq, tg, K)
        print("six K", K, "hbm status", g["status"], "oracle", w["status"])
else:
    st, v, sol = ctx.six_batch(0, True, tg[None], leq[None]); ctx.sync()
    w = port.six_solve(0, True, tg, gen.vc_nonneg(6, False).astype(np.float64), None, leq)
    print("six batch (LDS) status", st.tolist(), "oracle maxm", w[0])
