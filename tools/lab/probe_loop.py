"""GPU probe: per-call timing of DeviceLP.iterate to separate fixed stalls from per-pivot cost.
usage: probe_loop.py [torch] [sync]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
use_torch = "torch" in sys.argv
if use_torch:
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(4, device="cuda")
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, 0, leq, tg)
lp.begin()
lp.iterate(50)
ctx.sync()
if use_torch and "sync" in sys.argv:
    torch.cuda.synchronize()
for rep, (k, ev) in enumerate([(300, 0), (300, 300), (1000, 0), (1000, 1000), (300, 0)]):
    ctx.profile_begin(ev)
    t0 = time.perf_counter()
    st = lp.iterate(k)
    ctx.sync()
    dt = time.perf_counter() - t0
    n, ms = ctx.profile_end()
    print("iterate(%d) events=%d: %.1f us/pivot (status %d, sweep avg %.1f us)" % (k, ev, dt / k * 1e6, st, ms / max(n, 1) * 1e3))
