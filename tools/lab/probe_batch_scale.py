"""GPU probe: throughput of the LDS batch kernel against the batch size (pivots/s as well as LPs/s), both cfg-3 families."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
dev = torch.device("cuda", 0)
for fam in (1, 0):
    for nb in ([int(a) for a in sys.argv[1:]] or [1280, 2560, 5120, 8192, 16384, 32768, 65536]):
        leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
        d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
        d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
        d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        piv = d_piv.cpu().numpy().astype(np.int64)
        print("family %d nb %6d: %7.2f ms  %8.0f LPs/s  %6.1f M pivots/s  mean %5.0f max %5d pivots per LP" % (fam, nb, best * 1e3, nb / best, piv.sum() / best / 1e6, piv.mean(), piv.max()))
