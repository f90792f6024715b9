"""GPU probe (a -DXPG_STAMPS build): where a pivot of the LDS batch kernel's overlapped loop goes, per family."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from xpoly_amd._capi import lib
from tools import gen
ctx = xpoly_amd.Context(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
out = (C.c_ulonglong * 16)()
for fam in (1, 0):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
    d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    def run():
        ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
        ctx.sync()
    run()
    lib().xpg_fastloop_debug(ctx._h, out)
    t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
    lib().xpg_fastloop_debug(ctx._h, out)
    v = [int(x) for x in out]
    piv = int(d_piv.sum().item())
    it = v[0] + v[1]
    print("family %d: %d LPs %.1f ms, %d pivots; loop iterations %d: direct %d, through findpair %d (%.2f candidates tried each)" % (fam, nb, dt * 1e3, piv, it, v[0], v[1], v[2] / max(v[1], 1)))
    us = lambda k, n: v[k] / 100.0 / max(n, 1)
    print("   per iteration (us, wave 0 / thread 0): whole %.2f | findpair %.2f (per findpair) | stage A w0 %.2f, sweeper %.2f | stage C w0 %.2f, sweeper %.2f" % (us(7, it), us(3, v[1]), us(4, it), us(15, it), us(5, it), us(6, it)))
    print("   exits by action (mod 7):", v[8:15])
