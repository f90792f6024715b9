"""GPU probe (a -DXPG_LIFE build): how fast an LP of the dependence-test family pivots over its life, alone on its CU
and with 1..4 same-age siblings (all started together) -> where the lock-step slowdown sits."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from xpoly_amd._capi import lib
from tools import gen
ctx = xpoly_amd.Context(0)
dev = torch.device("cuda", 0)
fam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = lib()
L.xpg_life_debug.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros(4096 * 32, dtype=np.uint64)
for nb in ([int(a) for a in sys.argv[2:]] or [768]):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d_leq = torch.from_numpy(leq).to(dev); d_tg = torch.from_numpy(tg).to(dev)
    d_st = torch.empty(nb, dtype=torch.int32, device=dev); d_v = torch.empty(nb, dtype=torch.float64, device=dev)
    d_sol = torch.zeros(nb, 64, dtype=torch.float64, device=dev); d_piv = torch.empty(nb, dtype=torch.int32, device=dev)
    for rep in range(2):
        ctx.six_batch_dev(0, True, nb, d_tg.data_ptr(), d_leq.data_ptr(), 32, 64, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), d_piv.data_ptr())
        ctx.sync()
        assert L.xpg_life_debug(ctx._h, buf.ctypes.data_as(C.c_void_p)) == 0
    nbm = min(nb, 4096)                                                 # the marks cover the first 4096 LPs
    t = buf.reshape(4096, 32)[:nbm].astype(np.float64) / 100.0         # us
    piv = d_piv.cpu().numpy()[:nbm]
    nb_all, nb = nb, nbm
    t0 = t[:, 0].min()
    seg = []
    for k in range(1, 24):
        ok = (t[:, k] > 0) & (t[:, k - 1] > 0)
        seg.append((t[ok, k] - t[ok, k - 1]).mean() / 256.0 if ok.sum() > nb // 16 else float("nan"))
    # head: start -> first mark (pivot 256); tail: last mark -> end, against the pivots left after that mark
    last_k = np.array([max([k for k in range(1, 31) if t[b, k] > 0] or [0]) for b in range(nb)])
    tail = t[np.arange(nb), 31] - t[np.arange(nb), last_k]
    left = piv - last_k * 256
    head = np.where(t[:, 1] > 0, t[:, 1] - t[:, 0], np.nan)
    print("        head (256 pivots) mean %.0f us; tail after the last mark: mean %.0f us for %.0f pivots (%.2f us per pivot), max %.0f us; whole LP mean %.2f ms"
          % (np.nanmean(head), tail.mean(), left.mean(), tail.sum() / max(1, left.sum()), tail.max(), (t[:, 31] - t[:, 0]).mean() / 1e3))
    dur = (t[:, 31] - t[:, 0]) / 1e3
    order = np.argsort(-dur)[:6]
    marks = np.array([(t[b, 1:31] > 0).sum() for b in range(nb)])
    print("        LP durations (ms): p50 %.2f p90 %.2f p99 %.2f max %.2f; slowest: " % (np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max())
          + ", ".join("%.1f ms / %d pivots / %d marks / status %d" % (dur[b], piv[b], marks[b], int(d_st[b].item())) for b in order))
    b = int(order[0])
    print("        slowest LP %d, us per pivot per segment: " % b + " ".join("%.2f" % ((t[b, k] - t[b, k - 1]) / 256.0) for k in range(1, 31) if t[b, k] > 0 and t[b, k - 1] > 0)
          + " | tail %.0f us for %d pivots" % (tail[b], left[b]))
    started = t[:, 0] - t0
    print("nb %5d: LP start spread %.1f us, end %.2f ms; us per pivot over successive 256-pivot segments:" % (nb, started.max(), (t[:, 31].max() - t0) / 1e3))
    print("        " + " ".join("%.2f" % s for s in seg))
