"""GPU probe: pivot-count distribution of the dependence-test-like family and the per-pivot latency of a lone LP per CU."""
import os, sys, time
import numpy as np, torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
dev = torch.device("cuda", 0)
for fam in (0, 1):
    for nb in (8192, 256):
        leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
        d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
        d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
        d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
        def run():
            ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
            ctx.sync()
        run()
        t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
        piv = d_piv.cpu().numpy().astype(np.int64)
        q = np.percentile(piv, [50, 90, 99, 100]).astype(int).tolist()
        print("fam %d nb %5d: %.2f ms, pivots/LP median %d p90 %d p99 %d max %d, total %.1f M; max-LP alone would need %.1f us per pivot to fill the time"
              % (fam, nb, dt * 1e3, q[0], q[1], q[2], q[3], piv.sum() / 1e6, dt * 1e6 / max(1, q[3])))
