"""GPU probe: batched 32x64 LP throughput (device-resident inputs)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
for fam in (0, 1):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
    d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    def run():
        ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
        ctx.sync()
    run()
    t0 = time.perf_counter()
    for _ in range(3): run()
    dt = (time.perf_counter() - t0) / 3
    piv = int(d_piv.sum().item())
    print("fam", fam, "threads", os.environ.get("XPG_BATCH_THREADS", "default"), "LPs/s %.0f" % (nb / dt), "pivots/s %.1fM" % (piv / dt / 1e6), "ms %.2f" % (dt * 1e3), "checksum", int(d_st.sum().item()), float(d_v.sum().item()))
