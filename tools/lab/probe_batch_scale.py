"""GPU probe: throughput of the LDS batch kernel against the batch size (pivots/s as well as LPs/s), both cfg-3 families --
and what a launch of that size CAN reach: the LPs are dispatched in index order onto `slots` LDS slots (one workgroup per
LP, 5 per CU), so with the per-LP pivot counts the kernel reports, a greedy list schedule at the steady per-slot pivot rate
(taken from the largest batch of the run) predicts the launch time. Round 4: the "cold round at a third of the steady state"
of DESIGN section 7.3 is this schedule's tail (LP lengths are bimodal), not a collision inside the CU."""
import heapq
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
dev = torch.device("cuda", 0)
def list_schedule(piv, slots, us_per_pivot, us_fixed):
    """makespan (ms) of LPs taken in index order by whichever of `slots` frees first"""
    h = [0.0] * min(slots, len(piv))
    heapq.heapify(h)
    end = 0.0
    for p in piv:
        t = heapq.heappop(h) + us_fixed + float(p) * us_per_pivot
        end = max(end, t)
        heapq.heappush(h, t)
    return end / 1e3

SLOTS = 256 * 5
rows = {}
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
        rows.setdefault(fam, []).append((nb, best, piv))
        print("family %d nb %6d: %7.2f ms  %8.0f LPs/s  %6.1f M pivots/s  mean %5.0f max %5d pivots per LP" % (fam, nb, best * 1e3, nb / best, piv.sum() / best / 1e6, piv.mean(), piv.max()), flush=True)
for fam, rs in rows.items():
    nb, best, piv = max(rs, key=lambda r: r[0])
    # steady per-slot rate from the largest batch: all slots busy nearly all the time there
    us_per_pivot = best * 1e6 * SLOTS / max(1, piv.sum() + 40 * len(piv))      # (~40 pivot-times of set-up per LP: load, build, read-out)
    print("family %d: steady state %.2f us per pivot and slot (from %d LPs)" % (fam, us_per_pivot, nb))
    for nb, best, piv in rs:
        pred = list_schedule(piv, SLOTS, us_per_pivot, 40 * us_per_pivot)
        print("   nb %6d: measured %7.2f ms, list schedule of these LP lengths at the steady rate %7.2f ms (%.2f); all slots busy would be %7.2f ms" %
              (nb, best * 1e3, pred, best * 1e3 / pred, (piv.sum() + 40 * len(piv)) * us_per_pivot / SLOTS / 1e3))
