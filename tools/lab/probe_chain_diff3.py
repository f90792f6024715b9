"""GPU diagnostic (stamped build): after the first batch whose pivots differ, compare the staged rows E and
columns K of the chain-kernel loop with the launch-per-stage loop's."""
import ctypes as C
import os
import sys

import numpy as np

import xpoly_amd
from tools import gen
from xpoly_amd._capi import lib

F64 = 0
m, n = 300, 300
upto = int(sys.argv[1]) if len(sys.argv) > 1 else 352
leq, tg = gen.hard_lp_f64(m, n)
os.environ["XPG_LOOP"] = "block"
out = {}
for name, ch in (("chain", "1"), ("launch", "0")):
    os.environ["XPG_CHAIN"] = ch
    c = xpoly_amd.Context(0)
    lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
    lp.begin()
    for _ in range(upto // 16):
        lp.iterate(16)
    rc0 = np.zeros(600, dtype=np.int32); cc0 = np.zeros(600, dtype=np.int32)
    lib().xpg_lp_debug_counts(lp._h, rc0.ctypes.data_as(C.c_void_p), cc0.ctypes.data_as(C.c_void_p), C.c_int(600))
    print(name, "before the batch: rowcnt sum %d max %d, colcnt sum %d max %d, pivots %d" % (rc0.sum(), rc0.max(), cc0.sum(), cc0.max(), lp.pivots_done()))
    if name == "chain": cc_chain0 = cc0.copy(); rc_chain0 = rc0.copy()
    else:
        d = np.nonzero(cc0 != cc_chain0)[0]; print("colcnt before the batch differs at", d[:10].tolist(), cc_chain0[d[:10]].tolist(), cc0[d[:10]].tolist())
        d = np.nonzero(rc0 != rc_chain0)[0]; print("rowcnt before the batch differs at", d[:10].tolist(), rc_chain0[d[:10]].tolist(), rc0[d[:10]].tolist())
    lp.iterate(16)
    ld = C.c_int()
    lib().xpg_lp_debug_staged(lp._h, None, None, C.byref(ld))
    E = np.zeros((16, ld.value)); K = np.zeros((m, 16))
    lib().xpg_lp_debug_staged(lp._h, E.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), C.byref(ld))
    R = np.zeros((4, 8192))
    lib().xpg_lp_debug_rows(lp._h, R.ctypes.data_as(C.c_void_p))
    out[name] = (E, K, lp.trace()[-16:], lp.read(), R)
Ea, Ka, ta, ra, Ra = out["chain"]; Eb, Kb, tb, rb, Rb = out["launch"]
for q, nm in enumerate(("bc", "bi", "pair word", "colcnt")):
    d = np.nonzero(Ra[q, :m].view(np.uint64) != Rb[q, :m].view(np.uint64))[0]
    print("last pick, %s differs in rows %s" % (nm, d[:12].tolist()), [(Ra[q, i], Rb[q, i]) for i in d[:6]])
W = m + n + 1
print("pivots chain :", ta.tolist()); print("pivots launch:", tb.tolist())
for s in range(16):
    de = np.nonzero(Ea[s, :W].view(np.uint64) != Eb[s, :W].view(np.uint64))[0]
    dk = np.nonzero(Ka[:, s].view(np.uint64) != Kb[:, s].view(np.uint64))[0]
    print("stage %2d: E differs in %d columns %s   K differs in %d rows %s" % (s, len(de), de[:8].tolist(), len(dk), dk[:8].tolist()))
r = 99
print("row 99 K chain :", Ka[99].tolist()); print("row 99 K launch:", Kb[99].tolist())
