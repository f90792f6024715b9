import sys, time
sys.path.insert(0,'/root/repo')
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, 0, leq, tg)
t0=time.perf_counter()
st = lp.two_stage(0xFFFFFFFF)
print("status", st, "pivots", lp.pivots_done(), "seconds", time.perf_counter()-t0)
