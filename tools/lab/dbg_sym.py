import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import xpoly_amd
from xpoly_amd.six import has_solution
from oracle.checker import Port
from tools import gen
ctx = xpoly_amd.Context(0); port = Port()
rng = np.random.default_rng(4242)
nv, ns, rows = 3, 2, 9
mats = np.stack([gen.random_system(rng, rows, nv + ns) for _ in range(40)]); mats[..., 1] = 1
wide = np.zeros((nv + ns, nv + ns + 1), dtype=np.int32); wide[np.arange(nv), np.arange(nv)] = -1
wide = gen.to_rat(wide)
for b in range(40):
    moved = port.move2var(mats[b], nv, nv + 1, nv + ns)
    ok, res = port.reduce(moved, nv + ns, True)
    if not ok or res.shape[0] == 0: continue
    h = port.has_solution(res, None, wide, nv + ns, True, True)
    g = has_solution(ctx, res, None, wide, nv + ns, True, True)
    tg = np.zeros((nv+ns+1,), dtype=np.int32); tg[:nv+ns] = 1
    pm = [port.mip_solve(1, mx, False, gen.to_rat(tg), wide, None, res)[0] for mx in (True, False)]
    mip = xpoly_amd.six.MIP(ctx, 1)
    gm = []
    for mx in (True, False):
        try: gm.append((mip.maxm if mx else mip.minm)(gen.to_rat(tg), wide, None, res)[0])
        except Exception as e: gm.append(str(e)[:40])
    print(b, "rows", res.shape[0], "port has_sol", h, "gpu", g, "port mip max/min", pm, "gpu mip", gm)
