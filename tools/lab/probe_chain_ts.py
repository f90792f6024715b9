"""GPU diagnostic (stamped build): raw timeline of one chain launch at 4096 x 8192 (XPG_TS_M / XPG_TS_N / XPG_TS_KIND=dense: another LP) -- where the ~7 us of a
stage go. Worker classes: 0 = worker 0, 1 = last picker (63), 2 = first prep-only worker (64), 3 = last prepper."""
import ctypes as C

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd._capi import lib

ctx = xpoly_amd.Context(0)
import os
M_, N_ = int(os.environ.get("XPG_TS_M", "4096")), int(os.environ.get("XPG_TS_N", "4095"))
leq, tg = (gen.dense_lp_f64 if os.environ.get("XPG_TS_KIND") == "dense" else gen.hard_lp_f64)(M_, N_)
os.environ.setdefault("XPG_LOOP", "block")
lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
lp.begin(); lp.iterate(1600)
cap = C.c_int(0)
lib().xpg_lp_debug_chain_ts(lp._h, None, C.byref(cap), 0)          # the library's stage capacity (BLK_MAX)
ts = np.zeros((4, cap.value, 8), dtype=np.uint64)
assert lib().xpg_lp_debug_chain_ts(lp._h, ts.ctypes.data_as(C.c_void_p), None, cap.value) == 0
t = ts.astype(np.int64)
B = cap.value
nb = int(__import__("os").environ.get("XPG_BLOCK", "32"))
base = t[0, 1, 0]
pick_names = [(0, "top"), (1, "partials seen"), (2, "gather in"), (5, "replayed"), (6, "ratio+key"), (7, "wave min"), (3, "record out")]   # (the pick role no longer polls the records: the commit granule names the row)
prep_names = [(0, "top"), (4, "records seen"), (1, "row+payload in"), (2, "replayed"), (5, "scaled"), (3, "obj+pricing"), (6, "partial out")]
for wc, nm in enumerate(["pick worker 0", "last pick worker", "first prep worker", "last prep worker"]):
    names = pick_names if wc < 2 else prep_names
    print(nm)
    for st in (1, 2, 8, 9, 16, 17, nb - 2, nb - 1):
        row = [((t[wc, st, p] - base) * 0.01 if t[wc, st, p] else float("nan")) for p, _ in names]
        print("  stage %2d: " % st + "  ".join("%s %7.2f" % (n[:14], x) for (_, n), x in zip(names, row)))
    # mean segment lengths over the stages 2 .. nb-1 (us)
    seg = []
    for a, b in zip(names[:-1], names[1:]):
        d = (t[wc, 2:nb, b[0]] - t[wc, 2:nb, a[0]]) * 0.01
        seg.append("%s->%s %.2f" % (a[1], b[1], d.mean()))
    print("  mean segments (stages 2..%d): " % (nb - 1) + " | ".join(seg))
print("stage-to-stage (pick worker 0 top):", np.diff(t[0, 1:nb, 0]) * 0.01)
print("mean stage: %.3f us" % (np.diff(t[0, 1:nb, 0]).mean() * 0.01))
# the shader clock during the launch: s_memtime ticks per 100 MHz tick between the tops of consecutive stages (pick worker 0)
dc = np.diff(t[0, 1:nb, 4]).astype(float); dw = np.diff(t[0, 1:nb, 0]).astype(float)
if dc.sum() > 0: print("shader clock during the chain launch: %.0f MHz (min %.0f, max %.0f over the stages)" % ((dc.sum() / dw.sum()) * 100.0, (dc / dw).min() * 100.0, (dc / dw).max() * 100.0))
