"""GPU parity of the branch-and-bound controller (MIP<Mat,T>) and of
Lineq::has_solution: host DFS in the library, every node LP on the GPU; compared
with the golden vectors of the real reference and with the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
F64, RAT = 0, 1
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dec(d):
    return np.array(d["data"], dtype=np.int32).reshape(tuple(d["shape"]))


def test_mip_golden(ctx):
    import xpoly_amd
    from xpoly_amd.six import MIP
    g = json.load(open(os.path.join(GOLD, "g6_mip.json")))
    mip = MIP(ctx, RAT)
    n = 0
    for case in g:
        p = {k: dec(v) for k, v in case["problem"].items()}
        ind = np.array(case["ind"], dtype=np.uint8) if "ind" in case else None
        for key, is_max in (("max", True), ("min", False)):
            if key not in case:
                continue
            st, v, sol = (mip.maxm if is_max else mip.minm)(p["tgtf"], p["vc"], None, p["leq"], case["is_bin"], ind)
            w = case[key]
            assert st == w["status"], (n, key)
            assert v.tolist() == w["v"], (n, key)
            if st == 0:
                assert sol.reshape(-1).tolist() == w["sol"], (n, key)
            n += 1
    assert n > 40


@pytest.mark.parametrize("kind", [RAT, F64])
def test_mip_matches_oracle(ctx, port, kind):
    from xpoly_amd.six import MIP
    rng = np.random.default_rng(17 + kind)
    mip = MIP(ctx, kind)
    seen = set()
    for it in range(30):
        m, nv = int(rng.integers(1, 6)), int(rng.integers(1, 7))
        is_bin = bool(rng.integers(0, 2))
        p = gen.random_mip(rng, m, nv, is_bin)
        if kind == F64:
            p = {k: (v[..., 0].astype(np.float64) if k != "ind" else v) for k, v in p.items()}
        for is_max in (True, False):
            want = port.mip_solve(kind, is_max, is_bin, p["tgtf"], p["vc"], None, p["leq"], p.get("ind"))
            if want[0] == -7:
                continue
            got = (mip.maxm if is_max else mip.minm)(p["tgtf"], p["vc"], None, p["leq"], is_bin, p.get("ind"))
            assert got[0] == want[0], (it, is_max)
            assert np.array_equal(np.asarray(got[1]), np.asarray(want[1])), (it, is_max)
            if want[0] == 0:
                assert np.array_equal(got[2], want[2]), (it, is_max)
            seen.add(want[0])
    assert 0 in seen


def test_has_solution_golden_and_oracle(ctx, port):
    from xpoly_amd.six import has_solution
    g = json.load(open(os.path.join(GOLD, "g5_lineq.json")))
    seen = set()
    for c in g["has_solution"]:
        leq = dec(c["leq"])
        nv = leq.shape[1] - 1
        vc = gen.to_rat(gen.vc_nonneg(nv, False))
        r = has_solution(ctx, leq, None, vc, nv, c["is_int"], c["is_unique"])
        assert r == c["result"]
        seen.add(r)
    assert seen == {0, 1}
    rng = np.random.default_rng(3)
    for it in range(25):
        sysm, vc = gen.random_feas(rng, int(rng.integers(1, 9)), int(rng.integers(1, 6)))
        for ii in (True, False):
            for uu in (True, False):
                want = port.has_solution(sysm, None, vc, sysm.shape[1] - 1, ii, uu)
                if want == -7:
                    continue
                assert has_solution(ctx, sysm, None, vc, sysm.shape[1] - 1, ii, uu) == want, (it, ii, uu)


def test_mip_batch_lockstep_matches_oracle(ctx, port):
    """Many trees advanced together (node LPs of equal shape share a launch) give, per problem,
    exactly what the one-at-a-time recursion gives."""
    from xpoly_amd.six import mip_batch
    rng = np.random.default_rng(11)
    for is_bin in (False, True):
        for (m, nv) in ((3, 4), (2, 6)):
            nb = 40
            probs = [gen.random_mip(rng, m, nv, False) for _ in range(nb)]
            if is_bin:      # x <= 1 rows make it a 0-1 knapsack; keep rows <= cols so the reference is defined
                probs = [gen.random_mip(rng, 1, nv, False) for _ in range(nb)]
                for p in probs:
                    ub = np.zeros((nv, nv + 1), dtype=np.int32); ub[np.arange(nv), np.arange(nv)] = 1; ub[:, nv] = 1
                    p["leq"] = np.concatenate([p["leq"], gen.to_rat(ub)], axis=0)
            leq = np.stack([p["leq"] for p in probs]); tg = np.stack([p["tgtf"] for p in probs])
            for is_max in (True, False):
                st, v, sol, nodes = mip_batch(ctx, is_max, is_bin, tg, leq)
                assert nodes >= nb
                for b in range(nb):
                    want = port.mip_solve(RAT, is_max, is_bin, probs[b]["tgtf"], probs[b]["vc"], None, probs[b]["leq"])
                    assert st[b] == want[0], (is_bin, m, nv, is_max, b)
                    assert np.array_equal(v[b], want[1]), (is_bin, m, nv, is_max, b)
                    if want[0] == 0:
                        assert np.array_equal(sol[b], want[2]), (is_bin, m, nv, is_max, b)


def test_dep_is_empty_batch_matches_oracle(ctx, port):
    """DepPoly::is_empty composed from the oracle's reduce + has_solution(int, unique)."""
    from xpoly_amd.six import dep_is_empty_batch
    rng = np.random.default_rng(21)
    seen = set()
    for (rows, nv) in ((4, 2), (8, 3), (12, 4)):
        nb = 64
        mats = np.stack([gen.random_system(rng, rows, nv) for _ in range(nb)])
        mats[..., 1] = 1                                   # dependence polyhedra are integer systems
        empty, nodes = dep_is_empty_batch(ctx, mats)
        vc = gen.to_rat(gen.vc_nonneg(nv, False))
        for b in range(nb):
            ok, res = port.reduce(mats[b], nv, True)
            if not ok:
                want = 1
            elif res.shape[0] == 0:
                want = 0
            else:
                h = port.has_solution(res, None, vc, nv, True, True)
                want = h if h < 0 else int(not h)
            assert empty[b] == want, (rows, nv, b, empty[b], want)
            seen.add(want)
    assert {0, 1} <= seen


@pytest.mark.parametrize("kind", [F64, RAT])
def test_six_through_the_hbm_resident_path(ctx, port, kind, monkeypatch):
    """six.maxm/minm normally use the LDS kernel for small problems; force the HBM-resident
    loop (device LP + host dual construction) and check it against the oracle as well."""
    import xpoly_amd
    monkeypatch.setenv("XPG_FORCE_DEVICE_LP", "1")
    rng = np.random.default_rng(70 + kind)
    six = xpoly_amd.SIX(ctx, kind)
    for it in range(25):
        fam = int(rng.integers(0, 4))
        prob = gen.random_problem(rng, kind, fam, int(rng.integers(1, 8)), int(rng.integers(1, 8)))
        for is_max in (True, False):
            want = port.six_solve(kind, is_max, prob["tgtf"], prob["vc"], prob.get("eq"), prob.get("leq"))
            if want[0] == -7:
                continue
            got = (six.maxm if is_max else six.minm)(prob["tgtf"], prob["vc"], prob.get("eq"), prob.get("leq"))
            assert got[0] == want[0], (it, is_max)
            assert np.array_equal(np.asarray(got[1]), np.asarray(want[1]))
            if want[0] == 0:
                assert np.array_equal(got[2], want[2])


def test_mip_batch_host_controller_with_thread_pool_matches_the_device_tree_walk(ctx):
    """xpg_mip_batch_rat32 walks its trees on the device (one workgroup per problem); XPG_MIP_DEVICE=0 keeps the host
    controller (lock-step rounds), and XPG_HOST_THREADS > 1 hands that controller's per-tree loops to persistent
    host threads with static shares. Both knobs are read once per process, so that run is a child process; every
    status, value, solution and the node count must be the same either way."""
    import subprocess, sys, hashlib
    from xpoly_amd.six import mip_batch
    leq, tgtf = gen.knapsack_batch_rat(300, 12)
    st, v, sol, nodes = mip_batch(ctx, True, True, tgtf, leq)
    want = hashlib.sha256(st.tobytes() + v.tobytes() + sol.tobytes()).hexdigest() + " %d" % nodes
    code = ("import hashlib, xpoly_amd; from tools import gen; from xpoly_amd.six import mip_batch\n"
            "ctx = xpoly_amd.Context(0); leq, tgtf = gen.knapsack_batch_rat(300, 12)\n"
            "st, v, sol, nodes = mip_batch(ctx, True, True, tgtf, leq)\n"
            "print(hashlib.sha256(st.tobytes() + v.tobytes() + sol.tobytes()).hexdigest(), nodes)\n")
    env = dict(os.environ, XPG_HOST_THREADS="3", XPG_MIP_DEVICE="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == want


def test_mip_batch_f64_matches_oracle(ctx, port):
    """xpg_mip_batch_f64: the device tree walk in Float arithmetic (integer and 0-1 branching, maxm and minm)
    against MIP<FloatMat,Float> of the oracle, problem by problem."""
    from xpoly_amd.six import mip_batch
    rng = np.random.default_rng(23)
    checked = 0
    for is_bin in (False, True):
        for (m, nv) in ((3, 4), (2, 6), (4, 8)):
            nb = 48
            probs = [gen.random_mip(rng, 1 if is_bin else m, nv, False) for _ in range(nb)]
            if is_bin:
                for p in probs:
                    ub = np.zeros((nv, nv + 1), dtype=np.int32); ub[np.arange(nv), np.arange(nv)] = 1; ub[:, nv] = 1
                    p["leq"] = np.concatenate([p["leq"], gen.to_rat(ub)], axis=0)
            leq = np.stack([p["leq"][..., 0].astype(np.float64) for p in probs])
            tg = np.stack([p["tgtf"][..., 0].astype(np.float64) for p in probs])
            vc = probs[0]["vc"][..., 0].astype(np.float64)
            for is_max in (True, False):
                st, v, sol, nodes = mip_batch(ctx, is_max, is_bin, tg, leq, kind=F64)
                assert nodes >= nb
                for b in range(nb):
                    want = port.mip_solve(F64, is_max, is_bin, tg[b], vc, None, leq[b])
                    assert st[b] == want[0], (is_bin, m, nv, is_max, b, st[b], want[0])
                    assert np.array_equal(np.atleast_1d(v[b]), np.atleast_1d(want[1])), (is_bin, m, nv, is_max, b)
                    if want[0] == 0:
                        assert np.array_equal(sol[b], want[2]), (is_bin, m, nv, is_max, b)
                    checked += 1
    assert checked == 2 * 3 * 2 * 48


def test_mip_batch_bench_shape_matches_the_reference_fixture(ctx):
    """The device tree walk on 0-1 knapsacks of the bench shape (24 variables, 26 rows) against the answers of the
    REAL reference's MIP<RMat,Rational>::maxm(is_bin) (tests/golden/g10_mip_bench.json; inputs regenerated and
    checked by hash). Problems on which the reference is undefined return XPG_ERR_REF_UNDEFINED."""
    import hashlib
    from xpoly_amd.six import mip_batch
    g = json.load(open(os.path.join(GOLD, "g10_mip_bench.json")))
    leq, tgtf = gen.knapsack_batch_rat(g["nb"], g["nv"])
    assert hashlib.sha256(np.ascontiguousarray(leq).tobytes() + np.ascontiguousarray(tgtf).tobytes()).hexdigest() == g["inputs_sha256"]
    st, v, sol, nodes = mip_batch(ctx, True, True, tgtf, leq)
    for b in range(g["nb"]):
        want = g["results"][b]
        if want is None:
            assert st[b] == -7, b
            continue
        assert st[b] == want["status"], (b, st[b], want["status"])
        assert [int(v[b][0]), int(v[b][1])] == want["v"], b
        if want["status"] == 0:
            assert [int(x) for x in sol[b].reshape(-1)] == want["sol"], b


@pytest.mark.parametrize("kind", [RAT, F64])
def test_mip_with_root_equalities_on_the_device_matches_oracle(ctx, port, kind):
    """Equalities at the root (MIP::maxm's `eq`, what PolyTran::FeaSchedule passes) through the device tree walk:
    convertEq2Ineq's substitution of the root's and the branches' equalities per node, leftovers as pairs."""
    from mip_eq_cases import run
    compared, seen = run(ctx, port, kind, 4242 + kind, 120)
    print("MIPs with equalities compared:", compared, "status histogram:", seen)
    assert compared > 120 and 0 in seen and len(seen) >= 2


HOST_MIP_SCRIPT = r"""
import json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import xpoly_amd
from oracle.checker import Port
from mip_eq_cases import run
ctx = xpoly_amd.Context(0)
port = Port()
out = {}
for kind in (1, 0):
    compared, seen = run(ctx, port, kind, 4242 + kind, 60)
    out[str(kind)] = compared
print(json.dumps(out))
"""


def test_mip_with_root_equalities_on_the_host_controller_matches_oracle():
    """The same problems with XPG_MIP_DEVICE=0: the host controller (mip_host.hip.h run_mip_tasks) that still serves
    general variable constraints and problems beyond the LDS budget. The switch is read once per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XPG_MIP_DEVICE="0")
    r = subprocess.run([sys.executable, "-c", HOST_MIP_SCRIPT, root], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["1"] > 60 and out["0"] > 60


def test_has_solution_with_equalities_matches_oracle(ctx, port):
    """Lineq::has_solution (src/com/linsys.cpp:830-906) with an equality system beside the inequalities: the integer
    and the rational question, unique or not, on systems whose equalities pass through a lattice point or miss it.
    The integer question is MIP::maxm / minm with equalities at the root -- the device tree walk since round 3."""
    from xpoly_amd.six import has_solution
    rng = np.random.default_rng(909)
    seen = {}
    for it in range(80):
        nv = int(rng.integers(2, 6))
        leq, vc = gen.random_feas(rng, int(rng.integers(1, 7)), nv)
        xs = rng.integers(0, 4, size=nv)
        me = int(rng.integers(1, 3))
        Ae = rng.integers(-2, 3, size=(me, nv))
        be = Ae @ xs + (rng.integers(0, 2, size=me) if rng.random() < 0.25 else 0)
        eq = gen.to_rat(np.concatenate([Ae, np.asarray(be).reshape(me, 1)], axis=1).astype(np.int32))
        for ii in (True, False):
            for uu in (True, False):
                want = port.has_solution(leq, eq, vc, nv, ii, uu)
                if want == -7:
                    continue
                got = has_solution(ctx, leq, eq, vc, nv, ii, uu)
                assert got == want, (it, ii, uu, got, want)
                seen[want] = seen.get(want, 0) + 1
    print("has_solution with equalities:", seen)
    assert set(seen) == {0, 1}


def test_f64_mip_whose_nodes_pass_32_rows_in_a_block_carved_for_more(ctx, port):
    """25 inequalities, 35 integer variables, maximising: the tree's LDS block is carved for rmax = 60 rows with row
    stride 97, the walk runs 256 threads, and at depth 7 a node LP has exactly 32 live rows and stride 97 -- the shape
    the specialised pivot loop of the 32 x 64 batch kernel is compiled for, in a block laid out differently (its
    objective row sits 60 rows behind the tableau base, not 32). The trees must match the oracle's all the way down
    (round-3 advisor finding; lpsol.h:2427-2612)."""
    from xpoly_amd.six import MIP, mip_batch
    rng = np.random.default_rng(41)
    vc = gen.vc_nonneg(35, True)
    probs, wants, deep = [], [], 0
    while len(probs) < 12:
        leq, tg = gen.interval_mip_f64(rng, 25, 35, int(rng.choice([3, 6, 12])))
        st = {}
        w = port.mip_solve(F64, True, False, tg, vc, None, leq, stats=st)
        if w[0] < 0:
            continue
        deep += st["max_leq_rows"] >= 32
        probs.append((leq, tg)); wants.append(w)
    assert deep >= 6                                       # most of these trees do pass through 32 rows
    mip = MIP(ctx, F64)
    for (leq, tg), want in zip(probs, wants):              # one tree per call (xpg_mip_maxm_f64)
        got = mip.maxm(tg, vc, None, leq, False, None)
        assert got[0] == want[0]
        assert np.array_equal(np.atleast_1d(got[1]), np.atleast_1d(want[1]))
        if want[0] == 0:
            assert np.array_equal(got[2].reshape(-1), want[2].reshape(-1))
    st, v, sol, nodes = mip_batch(ctx, True, False, np.stack([p[1] for p in probs]), np.stack([p[0] for p in probs]), kind=F64)
    for b, want in enumerate(wants):                       # and as one batch (xpg_mip_batch_f64)
        assert st[b] == want[0], b
        assert np.array_equal(np.atleast_1d(v[b]), np.atleast_1d(want[1])), b
        if want[0] == 0:
            assert np.array_equal(sol[b].reshape(-1), want[2].reshape(-1)), b


@pytest.mark.parametrize("kind", [RAT, F64])
def test_mip_batch_with_root_equalities_matches_oracle(ctx, port, kind):
    """xpg_mip_batch_eq_*: batches of MIPs WITH equalities at the root (PolyTran::FeaSchedule's shape), every tree on the
    device, against MIP::maxm / minm of the oracle problem by problem -- integer and 0-1 branching, several shapes."""
    from mip_eq_cases import random_mip_eq
    from xpoly_amd.six import mip_batch_eq
    rng = np.random.default_rng(909 + kind)
    compared, seen = 0, set()
    for (m_leq, m_eq, nv) in ((3, 1, 4), (2, 2, 5), (0, 2, 3), (4, 3, 6)):
        for is_bin in (False, True):
            nb = 40
            probs = []
            for _ in range(nb):
                p = random_mip_eq(rng, m_leq, m_eq, nv, is_bin)
                p.pop("ind", None)
                probs.append(p)
            # one shape per call: the 0-1 family sometimes appends x <= 1 rows -- keep the problems of the commonest shape
            shapes = {}
            for p in probs:
                shapes.setdefault(None if p["leq"] is None else p["leq"].shape[0], []).append(p)
            probs = max(shapes.values(), key=len)
            conv = (lambda a: a[..., 0].astype(np.float64)) if kind == F64 else (lambda a: a)
            tg = np.stack([conv(p["tgtf"]) for p in probs]); eq = np.stack([conv(p["eq"]) for p in probs])
            leq = None if probs[0]["leq"] is None else np.stack([conv(p["leq"]) for p in probs])
            vc = conv(probs[0]["vc"])
            for is_max in (True, False):
                st, v, sol, nodes = mip_batch_eq(ctx, is_max, is_bin, tg, leq, eq, kind=kind)
                for b, p in enumerate(probs):
                    want = port.mip_solve(kind, is_max, is_bin, tg[b], vc, eq[b], None if leq is None else leq[b])
                    if want[0] == -7:
                        assert st[b] == -7, (m_leq, m_eq, nv, is_bin, is_max, b)
                        continue
                    assert st[b] == want[0], (m_leq, m_eq, nv, is_bin, is_max, b, st[b], want[0])
                    assert np.array_equal(np.atleast_1d(v[b]), np.atleast_1d(want[1])), (m_leq, m_eq, nv, is_bin, is_max, b)
                    if want[0] == 0:
                        assert np.array_equal(sol[b].reshape(-1), np.asarray(want[2]).reshape(-1)), (m_leq, m_eq, nv, is_bin, is_max, b)
                    compared += 1
                    seen.add(int(want[0]))
    assert compared > 200 and 0 in seen
