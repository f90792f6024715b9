"""GPU cross-check, larger than the collected test's (run by hand: python tests/crosscheck_mip_eq.py [count]): MIP::maxm /
minm with equalities at the root (tests/mip_eq_cases.py) against the oracle, rational and fp64, the device tree walk
(default) or, with XPG_MIP_DEVICE=0, the host controller."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import xpoly_amd
from oracle.checker import Port
from mip_eq_cases import run
ctx = xpoly_amd.Context(0); port = Port()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for kind in (1, 0):
    for seed in (11, 12, 13):
        compared, seen = run(ctx, port, kind, seed * 100 + kind, n)
        print("kind %d seed %d: %d compared, statuses %s, 0 mismatches" % (kind, seed, compared, seen))
