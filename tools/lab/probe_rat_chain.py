"""Where a fused Rational launch (lp_fused_r32.hip.h) spends its time: stamps of the pick -> stagers chain of the k-th
launch of the cfg-4 LP, k = 1..16 (a -DXPG_STAMPS build: XPG_SO_PATH=tools/_build/libxpoly_stamps.so)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import xpoly_amd
from xpoly_amd._capi import lib
from tools import gen

ctx = xpoly_amd.Context()
leq, tgtf = gen.int_lp_rat(1024, 1023)
lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.RAT, leq, tgtf)
L = lib()
print("launch: pick start -> rows done -> go -> [stager start, saw flag] -> column done -> descriptor   (us after the pick's start)")
for k in range(1, 17):
    lp.begin()
    lp.iterate(k)
    out = (C.c_ulonglong * 8)()
    L.xpg_lp_debug(lp._h, out)
    t = [int(x) for x in out]
    rel = [(x - t[0]) / 100.0 for x in t[:8]]
    print("%2d: rows %6.2f  record0 %6.2f  last record %6.2f  stager0 start %6.2f has all %6.2f  column %6.2f  desc %6.2f" % (k, rel[1], rel[7], rel[2], rel[3], rel[4], rel[5], rel[6]))
