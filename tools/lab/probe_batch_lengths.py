"""GPU probe: per-LP pivot counts of the two cfg-3 families (8192 LPs) -> gpurun_out/batch_lengths.npz"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
nb = 8192
dev = torch.device("cuda", 0)
out = {}
for fam in (0, 1):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
    d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
    ctx.sync()
    out["piv%d" % fam] = d_piv.cpu().numpy(); out["st%d" % fam] = d_st.cpu().numpy()
    p = out["piv%d" % fam]
    print("family", fam, "pivots: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" % (p.mean(), np.median(p), np.percentile(p, 90), np.percentile(p, 99), p.max()))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/batch_lengths.npz", **out)
