"""GPU probe: is a lone batch launch slower than the same launch inside a stream of launches (clock ramp after idle)?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import xpoly_amd
from tools import gen
ctx = xpoly_amd.Context(0)
dev = torch.device("cuda", 0)
def setup(nb, fam):
    leq, tg = gen.small_lp_batch_f64(nb, 32, 64, fam)
    d = dict(leq=torch.from_numpy(leq).to(dev), tg=torch.from_numpy(tg).to(dev), st=torch.empty(nb, dtype=torch.int32, device=dev),
             v=torch.empty(nb, dtype=torch.float64, device=dev), sol=torch.zeros(nb, 64, dtype=torch.float64, device=dev), piv=torch.empty(nb, dtype=torch.int32, device=dev), nb=nb)
    return d
def launch(d):
    ctx.six_batch_dev(0, True, d["nb"], d["tg"].data_ptr(), d["leq"].data_ptr(), 32, 64, d["st"].data_ptr(), d["v"].data_ptr(), d["sol"].data_ptr(), d["piv"].data_ptr())
for fam in (1, 0):
    small, big = setup(1280, fam), setup(8192, fam)
    for name, d in (("1280", small), ("8192", big)):
        launch(d); ctx.sync()
        time.sleep(0.5)
        t0 = time.perf_counter(); launch(d); ctx.sync(); lone = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(8): launch(d)
        ctx.sync(); stream = (time.perf_counter() - t0) / 8
        launch(big); launch(big); t0 = time.perf_counter(); ctx.sync(); t1 = time.perf_counter(); launch(d); ctx.sync(); warm = time.perf_counter() - t1
        print("family %d nb %s: lone after 0.5 s idle %7.2f ms, mean of 8 back-to-back %7.2f ms, right after two big launches %7.2f ms" % (fam, name, lone * 1e3, stream * 1e3, warm * 1e3))
