"""GPU diagnostic: where does the time of a pick / prep launch of the blocked loop go? Needs a library built
with -DXPG_STAMPS (tools/lab/run_stamps.sh builds tools/_build/libxpoly_stamps.so and points XPG_SO_PATH at it)."""
import ctypes as C
import os

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd._capi import lib

ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
lp.begin(); lp.iterate(3840)
lp.begin(); lp.iterate(3840)
d = (C.c_ulonglong * 8)()
lib().xpg_lp_debug(lp._h, d)
if os.environ.get("XPG_CHAIN", "1") != "0":
    names = ["partials seen -> records seen (pick segment)", "records seen -> partials seen (prep segment)"]
    for k in range(2):
        print("%-40s %6.2f us per chain stage" % (names[k], d[k] * 0.01 / (3840 * 15 / 16)))
    print("(launch-path stage 0 stamps are mixed into slots 0-7 as well: 1/16 of the pivots)")
    raise SystemExit
names = ["pick: state+partials+pre", "pick: round 2 + replay", "pick: division", "pick: arg-min",
         "prep: state+records+pre", "prep: round 2 + replay + objective", "prep: pricing partial", "prep: commit"]
for k in range(8):
    print("%-40s %6.2f us per pivot" % (names[k], d[k] * 0.01 / 3840))
print("pick body %.2f, prep body %.2f us (workgroup 0, lane 0)" % (sum(d[:4]) * 0.01 / 3840, sum(d[4:]) * 0.01 / 3840))
