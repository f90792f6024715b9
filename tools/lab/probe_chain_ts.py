"""GPU diagnostic (stamped build): raw timeline of one chain launch at 4096 x 8192 -- where the ~7 us of a
stage go. Worker classes: 0 = worker 0, 1 = last picker (63), 2 = first prep-only worker (64), 3 = last prepper."""
import ctypes as C

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd._capi import lib

ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
lp.begin(); lp.iterate(1600)
cap = C.c_int(0)
lib().xpg_lp_debug_chain_ts(lp._h, None, C.byref(cap), 0)          # the library's stage capacity (BLK_MAX)
ts = np.zeros((4, cap.value, 8), dtype=np.uint64)
assert lib().xpg_lp_debug_chain_ts(lp._h, ts.ctypes.data_as(C.c_void_p), None, cap.value) == 0
t = ts.astype(np.int64)
base = t[0, 1, 0]
names = ["top", "partials seen", "gather+fresh in", "record issued", "records seen", "row replayed", "partial issued"]
for wc, nm in enumerate(["worker 0", "last picker", "first prep-only", "last prepper"]):
    print(nm)
    for st in (1, 2, 8, 9, 14, 15):
        row = [(t[wc, st, p] - base) * 0.01 if t[wc, st, p] else float("nan") for p in range(7)]
        print("  stage %2d: " % st + "  ".join("%s %7.2f" % (names[p][:14], row[p]) for p in range(7)))
print("stage-to-stage (worker 0 top):", np.diff(t[0, 1:16, 0]) * 0.01)
