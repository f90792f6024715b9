"""GPU probe: batched rational (int32 num/den) LP throughput, dependence-test-like integer data, inputs resident."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from xpoly_amd import RAT
from tools import gen
ctx = xpoly_amd.Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
for (m, cols, fam) in ((12, 17, 1), (32, 64, 1), (32, 64, 0)):
    leq, tg = gen.small_lp_batch_f64(nb, m, cols, fam)
    if fam == 0:                                     # dense positive data as integers 1..9
        leq = np.floor(leq * 9.0) + 1.0; tg = np.floor(tg * 9.0) + (tg > 0)
    rl = gen.to_rat(leq.astype(np.int32)); rt = gen.to_rat(tg.astype(np.int32))
    d_leq = torch.from_numpy(rl).to(dev); d_tg = torch.from_numpy(rt).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, 2, dtype=torch.int32, device=dev)
    d_sol = torch.zeros(nb, cols, 2, dtype=torch.int32, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    def run():
        ctx.six_batch_dev(RAT, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), m, cols, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
        ctx.sync()
    run()
    t0 = time.perf_counter()
    for _ in range(3): run()
    dt = (time.perf_counter() - t0) / 3
    piv = int(d_piv.sum().item())
    print("rat %dx%d fam %d: LPs/s %.0f pivots/s %.2fM ms %.2f checksum %d %d %d" % (m, cols, fam, nb / dt, piv / dt / 1e6, dt * 1e3,
          int(d_st.sum().item()), int(d_v[:, 0].to(torch.int64).sum().item()), int(d_sol.to(torch.int64).sum().item())))
