"""GPU diagnostic (build with -DXPG_EXP_STAMPS, load via XPG_SO_PATH): per-phase cycles of a pivot."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
nb = 8192
dev = torch.device("cuda", 0)
leq, tg = gen.small_lp_batch_f64(nb, 32, 64, 0)
d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
ctx.sync()
s = d_sol.cpu().numpy(); piv = d_piv.cpu().numpy().astype(np.float64)
tot = piv.sum()
for w in range(4):
    sel, b1, pv, stg = (s[:, 8 * w + q].sum() / tot for q in range(4))
    print("wave %d: per pivot cycles: selection %.0f, wait at barrier-1 %.0f, pivot (stage+sweep+2 barriers) %.0f of which staging+barrier %.0f"
          % (w, sel, b1, pv, stg))
