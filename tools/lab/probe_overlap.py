"""GPU probe: how much does the pick -> prep latency chain of the blocked loop suffer when something
else streams HBM at the same time? (Decides whether hiding the sweep behind the next batch's chain --
an out-of-place sweep on a second stream -- can pay.)
  1. LP A alone: pivots/s of xpg_lp_iterate(3840).
  2. A while a second replica B runs the same loop on its own context/stream (HBM ~30 % busy).
  3. A while B runs back-to-back one-shot K1 sweeps on its own tableau (HBM saturated)."""
import threading
import time

import numpy as np

import xpoly_amd
from tools import gen

M, N = 4096, 4095
leq, tg = gen.hard_lp_f64(M, N)
ca, cb = xpoly_amd.Context(0), xpoly_amd.Context(0)
A = xpoly_amd.DeviceLP(ca, xpoly_amd.F64, leq, tg)
B = xpoly_amd.DeviceLP(cb, xpoly_amd.F64, leq, tg)


def run(lp, n=3840):
    lp.begin()
    t0 = time.perf_counter()
    lp.iterate(n)
    return n / (time.perf_counter() - t0)


for _ in range(3):
    run(A); run(B)
print("1. A alone: %.0f pivots/s" % np.median([run(A) for _ in range(5)]))

res = {}
def worker(name, lp):
    res[name] = [run(lp) for _ in range(5)]
ta = threading.Thread(target=worker, args=("A", A)); tb = threading.Thread(target=worker, args=("B", B))
ta.start(); tb.start(); ta.join(); tb.join()
print("2. two replicas at once: A %.0f, B %.0f pivots/s each" % (np.median(res["A"]), np.median(res["B"])))

# 3. B = one-shot K1 sweeps back to back on a tableau of its own
W = M + N + 1
tab, obj = gen.tableau_f64(M, W)
d_tab = cb.malloc(tab.nbytes); d_obj = cb.malloc(obj.nbytes)
cb.upload(d_tab, tab); cb.upload(d_obj, obj)
stop = [False]
def sweeps():
    k = 0
    while not stop[0]:
        for _ in range(64):
            cb.pivot_dev(xpoly_amd.F64, d_tab, M, W, W, d_obj, W - 1, (k * 37) % M, (k * 101) % (W - 1)); k += 1
        cb.sync()
tb = threading.Thread(target=sweeps); tb.start()
time.sleep(0.2)
r = [run(A) for _ in range(5)]
stop[0] = True; tb.join()
print("3. A under back-to-back K1 sweeps of another tableau: %.0f pivots/s" % np.median(r))
