"""GPU diagnostic (build with -DXPG_EXP_STAMPS, load via XPG_SO_PATH): cycles per phase of a pivot of
the register-resident batch loop, waves 0 and 3."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
nb = 8192
dev = torch.device("cuda", 0)
for fam in (0, 1):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
    d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
    ctx.sync()
    s = d_sol.cpu().numpy(); tot = d_piv.cpu().numpy().astype(np.float64).sum()
    names = ["pricing", "export+B2", "ratio+B3", "staging+B4", "sweep+obj+B5"]
    for w, off in ((0, 32), (3, 37)):
        print("family %d wave %d:" % (fam, w), ", ".join("%s %.0f" % (names[q], s[:, off + q].sum() / tot) for q in range(5)),
              " total %.0f cycles/pivot" % (s[:, off:off + 5].sum() / tot))
