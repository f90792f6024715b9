"""GPU probe: how predictable are a batch's entering columns at the END of the batch before it? For a compact copy of
candidate columns written by the sweep (so that the chain's column gathers hit the L2 instead of HBM): the candidates
would be the first C nonbasic columns with a positive reduced cost when the previous batch is committed; this prints,
for several C, the share of the next batch's entering columns among them, on the two bench LPs."""
import sys

import numpy as np

import xpoly_amd
from tools import gen

B = 24
ctx = xpoly_amd.Context(0)
for (m, n) in ((4096, 4095), (4096, 8192)):
    leq, tg = gen.hard_lp_f64(m, n)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tg)
    del leq
    lp.begin()
    nb = 50
    lists = []
    for b in range(nb + 1):
        s = lp.read(want_tab=False)
        obj, nv = s["tgtf"], s["nvset"]
        rhs = len(nv)
        cand = np.nonzero((nv != 0) & (obj[:rhs] > 0))[0]
        lists.append(cand)
        lp.iterate(B)
    tr = lp.trace()
    enter = tr[:, 0]
    print("LP %d x %d: %d pivots traced, distinct entering columns per batch of %d: %.1f, positive nonbasic columns: %s" % (
        m, n, len(enter), B, np.mean([len(set(enter[b * B:(b + 1) * B])) for b in range(nb)]), [len(c) for c in lists[:6]]))
    for Cn in (8, 16, 32, 64, 128, 256, 1024):
        hit = tot = 0
        for b in range(nb):
            c = set(lists[b][:Cn].tolist())
            e = enter[b * B:(b + 1) * B]
            hit += sum(1 for x in e if int(x) in c); tot += len(e)
        print("  first %4d candidates: %.1f %% of the next batch's entering columns" % (Cn, 100.0 * hit / tot))
    # and by rank: where in the candidate list of the batch start does each entering column sit?
    ranks = []
    for b in range(nb):
        pos = {int(x): k for k, x in enumerate(lists[b])}
        ranks += [pos.get(int(x), -1) for x in enter[b * B:(b + 1) * B]]
    ranks = np.array(ranks)
    print("  rank percentiles (of those present): ", np.percentile(ranks[ranks >= 0], [50, 75, 90, 95, 99]).astype(int), " absent: %.1f %%" % (100.0 * np.mean(ranks < 0)))
    print("  first 48 entering columns:", enter[:48].tolist())
    lp.close()
