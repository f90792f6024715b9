"""GPU cross-check (run by hand): SIX::pivot (xpg_pivot_f64 / xpg_pivot_rat32, lpsol.h:1456-1511) on tableaux with cells
outside ordinary arithmetic -- fp64 inf / NaN / -0.0 / denormals, rational n/0, 0/d, negative and unreduced denominators,
values at the appro threshold -- and pivot elements 0, 1, -1, against the oracle's pivot."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import xpoly_amd
from oracle.checker import Port
ctx = xpoly_amd.Context(0); port = Port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
bad = total = 0
def same_f(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
for it in range(300):
    m, W = int(rng.integers(1, 40)), int(rng.integers(2, 140))
    row, col = int(rng.integers(0, m)), int(rng.integers(0, W - 1))
    # fp64
    tab = rng.uniform(-3, 3, size=(m, W)); tab[rng.random((m, W)) < 0.2] = 0.0; obj = rng.uniform(-2, 2, size=W)
    for _ in range(int(rng.integers(0, 6))):
        tab[int(rng.integers(0, m)), int(rng.integers(0, W))] = rng.choice([np.inf, -np.inf, np.nan, -0.0, 5e-324, 1e308, -1e308])
    if rng.random() < 0.3: obj[int(rng.integers(0, W))] = rng.choice([np.inf, np.nan, -0.0, 1.0, 0.0])
    tab[row, col] = rng.choice([1.0, -1.0, 0.5, 3.0, 1e-300, 1e300, 0.0, np.inf]) if rng.random() < 0.5 else rng.uniform(0.2, 2.0)
    wt, wo = tab.copy(), obj.copy()
    port.lib.orc_pivot_f64(wt.ctypes.data_as(C.c_void_p), C.c_int(m), C.c_int(W), wo.ctypes.data_as(C.c_void_p), C.c_int(W - 1), C.c_int(row), C.c_int(col))
    gt_, go = ctx.pivot(0, tab.copy(), obj.copy(), W - 1, row, col)
    total += 1
    if not (same_f(np.asarray(gt_), wt) and same_f(np.asarray(go), wo)):
        bad += 1
        if bad <= 4:
            d = np.argwhere(~((np.asarray(gt_) == wt) | (np.isnan(gt_) & np.isnan(wt))))
            print("f64 MISMATCH it", it, m, W, "pivot", row, col, tab[row, col], "first diffs", d[:3].tolist(), [(float(np.asarray(gt_)[tuple(x)]), float(wt[tuple(x)]), float(tab[tuple(x)])) for x in d[:3]])
    # rational
    scale = int(rng.choice([5, 60, 4000, 3000000, 0x3fffffff]))
    rt = np.zeros((m, W, 2), dtype=np.int32)
    rt[..., 0] = rng.integers(-scale, scale + 1, size=(m, W)); rt[..., 1] = rng.integers(1, min(scale, 0x7ffffffe) + 1, size=(m, W))
    ro = np.zeros((W, 2), dtype=np.int32); ro[:, 0] = rng.integers(-9, 10, size=W); ro[:, 1] = rng.integers(1, 10, size=W)
    for _ in range(int(rng.integers(0, 6))):
        rt[int(rng.integers(0, m)), int(rng.integers(0, W))] = [(int(rng.integers(-3, 4)), 0), (0, 5), (3, -2), (6, 4), (0x7ffffffe, 1), (1, 0x7ffffffe)][int(rng.integers(0, 6))]
    if rt[row, col, 0] == 0: rt[row, col] = (3, 1)
    if rng.random() < 0.3: rt[row, col] = [(1, 1), (-1, 1), (2, 2), (1, 0), (0, 1)][int(rng.integers(0, 5))]
    wt, wo = rt.copy(), ro.copy()
    port.lib.orc_pivot_rat32(wt.ctypes.data_as(C.c_void_p), C.c_int(m), C.c_int(W), wo.ctypes.data_as(C.c_void_p), C.c_int(W - 1), C.c_int(row), C.c_int(col))
    gt_, go = ctx.pivot(1, rt.copy(), ro.copy(), W - 1, row, col)
    total += 1
    if not (np.array_equal(gt_, wt) and np.array_equal(go, wo)):
        bad += 1
        if bad <= 4:
            d = np.argwhere((np.asarray(gt_) != wt).any(axis=2))
            print("rat MISMATCH it", it, m, W, "pivot", row, col, rt[row, col].tolist(), "first diffs", d[:3].tolist(), [(np.asarray(gt_)[tuple(x)].tolist(), wt[tuple(x)].tolist(), rt[tuple(x)].tolist()) for x in d[:3]])
print("compared", total, "mismatches", bad)
