"""CPU check of the ORACLE against the REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile) on
inputs outside ordinary arithmetic: SIX with equalities whose substitution divides by zero, rational tableaux with
n/0 cells, fp64 tableaux with inf / NaN. The reference may die on such inputs (integer division by zero is a SIGFPE
there), so every case runs in a child process; a case the reference does not survive is skipped.
Run by hand in the build container: python tools/crosscheck_oracle_weird.py [cases]
                                    python tools/crosscheck_oracle_weird.py overflow [trials]   (finite LPs that overflow mid-solve)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tools import gen
from oracle.checker import Port, Ref
which = sys.argv[2]
L = Port() if which == "port" else Ref()
case = json.loads(sys.stdin.read())
kind = case["kind"]
def arr(x): return None if x is None else np.array(x, dtype=np.float64 if kind == 0 else np.int32)
leq, eq, tg, vc = arr(case["leq"]), arr(case.get("eq")), arr(case["tg"]), arr(case["vc"])
out = {}
def enc(a):
    a = np.asarray(a)
    if a.dtype == np.float64: return ["nan" if x != x else repr(float(x)) for x in a.reshape(-1)]
    return a.reshape(-1).tolist()
if case["what"] == "six":
    for is_max in (True, False):
        r = L.six_solve(kind, is_max, tg, vc, eq, leq)
        out["max" if is_max else "min"] = [int(r[0]), enc(r[1]), enc(r[2]) if r[0] == 0 else None]
elif case["what"] == "two_stage":
    for K in case["Ks"]:
        r = L.two_stage(kind, leq, tg, K)
        out[str(K)] = [int(r["status"])] + ([enc(r["tab"]), enc(r["tgtf"]), enc(r["eq2bv"])] if r["status"] not in (2, -7) else [])
elif case["what"] == "has_solution":
    for ii in (True, False):
        for uu in (True, False):
            out["%d%d" % (ii, uu)] = int(L.has_solution(leq, eq, vc, leq.shape[1] - 1, ii, uu))
print("RESULT " + json.dumps(out))
"""

def run(which, case):
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, which], input=json.dumps(case), capture_output=True, text=True, timeout=120)
    for ln in p.stdout.splitlines():
        if ln.startswith("RESULT "): return json.loads(ln[7:])
    return None                                              # died (SIGFPE / assert) or reported nothing

def overflow(trials):
    """gen.overflow_lp_f64: FINITE input whose products overflow after a few pivots (inf - inf -> NaN cells mid-solve, NaN
    ratios in findPivotBV whose outcome only the reference's scan order decides): restatement against the real reference at
    iteration limits around the first NaN and at the end -- the same LPs tests/test_gpu_edges.py runs through the blocked loop."""
    from tools import gen
    from oracle.checker import Port
    port = Port()
    compared = bad = cases = 0
    for trial in range(trials):
        leq, tg = gen.overflow_lp_f64(trial)
        if not np.isnan(port.two_stage(0, leq, tg, 0xFFFFFFFF)["tab"]).any(): continue
        lo = next(K for K in range(1, 1000) if np.isnan(port.two_stage(0, leq, tg, K)["tab"]).any())
        cases += 1
        case = dict(what="two_stage", kind=0, tg=tg.tolist(), vc=gen.vc_nonneg(leq.shape[1] - 1, True).tolist(), leq=leq.tolist(),
                    Ks=sorted({max(1, lo - 1), lo, lo + 1, lo + 2, lo + 5, 100000}))
        a = run("port", case); r = run("ref", case)
        if r is None or a is None: bad += 1; print("DIED", trial, a is None, r is None); continue
        for k in r:
            compared += 1
            if a.get(k) != r[k]:
                bad += 1
                if bad <= 5: print("MISMATCH trial", trial, "K", k, "\n oracle", str(a.get(k))[:200], "\n ref   ", str(r[k])[:200])
    print("overflow LPs", cases, "compared", compared, "mismatches", bad)

def main():
    if len(sys.argv) > 1 and sys.argv[1] == "overflow":
        return overflow(int(sys.argv[2]) if len(sys.argv) > 2 else 1200)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(20261003)
    from tools import gen
    compared = died = bad = undefined = 0
    for it in range(n):
        what = ["six", "two_stage", "has_solution", "two_stage"][it % 4]
        kind = 1 if what == "has_solution" else int(rng.integers(0, 2))
        nv = int(rng.integers(2, 6)); ml = int(rng.integers(1, 7)); me = int(rng.integers(1, 3))
        A = rng.integers(-3, 5, size=(ml, nv)); b = rng.integers(-3, 9, size=ml); c = rng.integers(-2, 6, size=nv)
        xs = rng.integers(0, 4, size=nv); Ae = rng.integers(-2, 3, size=(me, nv)); be = Ae @ xs
        leq = np.concatenate([A, b[:, None]], axis=1); eq = np.concatenate([Ae, be[:, None]], axis=1); tg = np.concatenate([c, [0]])
        vc = gen.vc_nonneg(nv, False)
        if kind == 1:
            leq, eq, tg, vc = (gen.to_rat(x.astype(np.int32)) for x in (leq, eq, tg, vc))
        else:
            leq, eq, tg, vc = (x.astype(np.float64) for x in (leq, eq, tg, vc))
        case = dict(what=what, kind=kind, tg=tg.tolist(), vc=vc.tolist())
        if what == "two_stage":
            for _ in range(int(rng.integers(1, 3))):
                i, j = int(rng.integers(0, ml)), int(rng.integers(0, nv + 1))
                if kind == 1: leq[i, j] = (int(rng.choice([-2, -1, 1, 3])), 0)
                else: leq[i, j] = float(rng.choice([np.inf, -np.inf]))       # (NaN literals do not survive JSON)
            case["Ks"] = [0, 1, 2, 3, 6, 1000]
            case["leq"] = [[("inf" if x == np.inf else "-inf" if x == -np.inf else x) for x in row] for row in leq.tolist()] if kind == 0 else leq.tolist()
            if kind == 0: case["leq"] = [[float(x) for x in row] for row in case["leq"]]
        else:
            case["leq"] = leq.tolist(); case["eq"] = eq.tolist()
        a = run("port", case); r = run("ref", case)
        if r is None: died += 1; continue
        if a is None: bad += 1; print("ORACLE DIED", case); continue
        for k in r:
            if a.get(k) is not None and a[k][0] == -7 if isinstance(a.get(k), list) else a.get(k) == -7: undefined += 1; continue
            compared += 1
            if a.get(k) != r[k]:
                bad += 1
                if bad <= 5: print("MISMATCH", what, "kind", kind, "key", k, "\n oracle", str(a.get(k))[:300], "\n ref   ", str(r[k])[:300], "\n case", json.dumps(case)[:600])
    print("compared", compared, "mismatches", bad, "reference died on", died, "cases; oracle says undefined for", undefined)

if __name__ == "__main__":
    main()
