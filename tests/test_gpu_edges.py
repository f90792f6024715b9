"""GPU edge cases: malformed shapes, empty / ragged inputs, size limits, the inputs on which the
x86-64 reference is undefined, iteration caps (SIX::set_param) and size-independent properties
at BASELINE.json's full sizes."""
import ctypes as C

import numpy as np
import pytest

from conftest import needs_hooks

from tools import gen

pytestmark = pytest.mark.gpu
F64, RAT = 0, 1


def test_shape_errors_are_negative_codes_not_crashes(ctx):
    from xpoly_amd._capi import lib, vp
    L = lib()
    t = np.zeros(4); vc = np.zeros((3, 4)); leq = np.zeros((2, 4)); v = np.zeros(1); sol = np.zeros(4)
    # no constraints at all (reference: ASSERT "no constraints", lpsol.h:1538)
    assert L.xpg_six_maxm_f64(ctx._h, vp(t), vp(vc), 3, None, 0, None, 0, 4, 10, vp(v), vp(sol)) == -3
    # vc with the wrong number of rows (lpsol.h:1555-1557)
    assert L.xpg_six_maxm_f64(ctx._h, vp(t), vp(vc), 2, None, 0, vp(leq), 2, 4, 10, vp(v), vp(sol)) == -3
    # pivot outside the tableau
    tab = np.ones((4, 8)); obj = np.ones(8)
    assert L.xpg_pivot_f64(ctx._h, vp(tab), 4, 8, vp(obj), 7, 9, 0) == -3
    assert L.xpg_pivot_f64(ctx._h, vp(tab), 4, 8, vp(obj), 7, 0, 8) == -3
    # null context
    assert L.xpg_sync(None) == -3
    # empty batches are fine
    st = np.zeros(1, dtype=np.int32)
    assert L.xpg_six_batch_f64(ctx._h, 1, 0, vp(t), vp(leq), 2, 4, 10, vp(st), vp(v), vp(sol)) == 0


def test_lds_limit_reports_unsupported(ctx):
    from xpoly_amd._capi import lib, vp
    m, cols = 200, 201          # 200 x 402 slack tableau does not fit one CU's LDS
    leq = np.ones((1, m, cols)); tg = np.ones((1, cols))
    st = np.zeros(1, dtype=np.int32); v = np.zeros(1); sol = np.zeros((1, cols))
    rc = lib().xpg_six_batch_f64(ctx._h, 1, 1, vp(tg), vp(leq), m, cols, 10, vp(st), vp(v), vp(sol))
    assert rc == -4
    # ... but SIX::maxm on the same problem falls back to the HBM-resident loop and works
    import xpoly_amd
    six = xpoly_amd.SIX(ctx, F64)
    leq2, tg2 = gen.dense_lp_f64(200, 200)
    six.set_param(0, 5)
    st, _, _ = six.maxm(tg2, gen.vc_nonneg(200), None, leq2)
    assert st in (0, 1, 3, 4)


def test_reference_undefined_inputs_are_refused_or_defined(ctx, port):
    import xpoly_amd
    six = xpoly_amd.SIX(ctx, RAT)
    # equality substitution whose row index leaves the equality row (lpsol.h:1232): refused
    nv = 2
    leq = gen.to_rat(np.array([[1, 1, 4], [1, 0, 3], [0, 1, 3], [1, 2, 9]], dtype=np.int32))
    eq = gen.to_rat(np.array([[1, -1, 0]], dtype=np.int32))
    st, _, _ = six.maxm(gen.to_rat(np.array([1, 1, 0], dtype=np.int32)), gen.to_rat(gen.vc_nonneg(nv, False)), eq, leq)
    want = port.six_solve(RAT, True, gen.to_rat(np.array([1, 1, 0], dtype=np.int32)), gen.to_rat(gen.vc_nonneg(nv, False)), eq, leq)
    assert st == want[0]
    # a free variable: the reference's vcmap is garbage (sete), we implement v = v' - v''
    vc = gen.vc_nonneg(2, True, free=(1,))
    six = xpoly_amd.SIX(ctx, F64)
    # max x0 - x1  s.t. x0 <= 3, x1 >= -2 (free), x0 - x1 <= 4  ->  optimum 4
    leq = np.array([[1, 0, 3], [0, -1, 2], [1, -1, 4]], dtype=np.float64)
    st, v, sol = six.maxm([1.0, -1.0, 0.0], vc, None, leq)
    assert st == 0 and v == 4.0
    assert abs((sol[0] - sol[1]) - 4.0) < 1e-12 and sol[2] == 1.0


@pytest.mark.parametrize("kind", [F64, RAT])
def test_max_iter_gives_time_out_and_exact_pivot_count(ctx, kind):
    """SIX::set_param(indent, max_iter): K pivots then SIX_TIME_OUT (lpsol.h:1039, :1190)."""
    import xpoly_amd
    leq, tg = (gen.hard_lp_f64(24, 23) if kind == F64 else gen.int_lp_rat(24, 23))
    for K in (0, 1, 7):
        lp = xpoly_amd.DeviceLP(ctx, kind, leq, tg)
        assert lp.two_stage(K) == 4
        assert lp.pivots_done() == K
        lp.close()


def test_iterate_in_chunks_equals_one_shot(ctx):
    """Idempotence of the device loop: 5+7+11 pivots == 23 pivots, bit for bit."""
    import xpoly_amd
    leq, tg = gen.hard_lp_f64(40, 39)
    a = xpoly_amd.DeviceLP(ctx, F64, leq, tg); a.begin()
    for k in (5, 7, 11):
        a.iterate(k)
    b = xpoly_amd.DeviceLP(ctx, F64, leq, tg); b.begin(); b.iterate(23)
    ra, rb = a.read(), b.read()
    assert np.array_equal(ra["tab"].view(np.uint64), rb["tab"].view(np.uint64))
    assert np.array_equal(a.trace(), b.trace()) and len(a.trace()) == 23
    a.close(); b.close()


def test_full_size_tableau_properties(ctx, port):
    """4096 x 8192 fp64 (BASELINE.json configs[1]): after K pivots of the device loop
    (a) every basic column is a unit vector up to the reference's own rounding,
    (b) the objective constant only grows (maximisation, nonnegative ratios),
    (c) a 64-row random sample of the tableau equals the oracle's rows bit for bit."""
    import xpoly_amd
    m, n = 4096, 4095
    leq, tg = gen.hard_lp_f64(m, n)
    lp = xpoly_amd.DeviceLP(ctx, F64, leq, tg)
    lp.begin()
    consts = []
    for _ in range(3):
        assert lp.iterate(4) == xpoly_amd.six.XPG_RUNNING
        consts.append(lp.read(want_tab=False)["tgtf"][-1])
    st = lp.read()
    lp.close()
    assert consts[0] <= consts[1] <= consts[2] and consts[2] > 0
    tab, eq2bv = st["tab"], st["eq2bv"]
    assert st["bvset"].sum() == m and st["nvset"].sum() == n
    for r in (0, 17, 4095):
        col = tab[:, eq2bv[r]]
        assert abs(col[r] - 1.0) < 1e-9 and np.abs(np.delete(col, r)).max() < 1e-9
    # oracle replay of the same 12 pivots on the CPU, compared on a row sample
    want = port.two_stage(F64, leq, tg, 12)
    rows = np.random.default_rng(0).integers(0, m, 64)
    assert np.array_equal(tab[rows].view(np.uint64), want["tab"][rows].view(np.uint64))
    assert np.array_equal(st["tgtf"].view(np.uint64), want["tgtf"].view(np.uint64))
    assert np.array_equal(eq2bv, want["eq2bv"])


@needs_hooks
def test_full_size_loops_against_each_other_and_oracle(port, monkeypatch):
    """4096 x 8192 fp64 under full load: the three device loops -- blocked (16 pivots staged per
    sweep, the default at this size), pipelined (pick workgroups inside the sweep launch) and the
    serial three-launch loop -- and the CPU oracle must agree bit for bit: same (entering, leaving)
    trace, same basis, same tableau -- ALL 4096 x 8192 entries, between the GPU loops after 300 and
    after 2000 pivots and against the oracle after 300 (the chunks 7 + 93 + 200 also end batches early,
    so the short-batch sweep kernels are on the path)."""
    import xpoly_amd
    m, n, K = 4096, 4095, 300
    leq, tg = gen.hard_lp_f64(m, n)
    got = {}
    for mode in ("block", "pipe", "serial"):
        monkeypatch.setenv("XPG_LOOP", mode)             # read when the context is created
        c = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
        lp.begin()
        for k in (7, 93, 200):                           # uneven chunks: 300 pivots in all
            assert lp.iterate(k) == xpoly_amd.six.XPG_RUNNING
        got[mode] = (lp.read(), lp.trace().copy(), lp.pivots_done())
        assert lp.iterate(1700) == xpoly_amd.six.XPG_RUNNING     # and on to 2000 pivots, GPU loops only
        got[mode + "+"] = (lp.read(), lp.trace().copy(), lp.pivots_done())
        lp.close(); c.close()
    for tag, total in (("", K), ("+", 2000)):
        b, tb, nb = got["serial" + tag]
        for mode in ("block", "pipe"):
            a, ta, na = got[mode + tag]
            assert na == nb == total and np.array_equal(ta, tb), (mode, tag)
            for k in ("tab", "tgtf"):
                assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), (mode, tag, k)
            for k in ("nvset", "bvset", "bv2eq", "eq2bv"):
                assert np.array_equal(a[k], b[k]), (mode, tag, k)
    a, ta, _ = got["block"]
    want = port.two_stage(F64, leq, tg, K)               # ~5 s of CPU: the whole tableau, not a sample
    assert a["tab"].shape == want["tab"].shape == (m, m + n + 1)
    assert np.array_equal(a["tab"].view(np.uint64), want["tab"].view(np.uint64))
    assert np.array_equal(a["tgtf"].view(np.uint64), want["tgtf"].view(np.uint64))
    for k in ("nvset", "bvset", "bv2eq", "eq2bv"):
        assert np.array_equal(a[k], want[k]), k


def test_midsize_whole_solve_every_loop(port, monkeypatch):
    """A 300 x 300 fp64 LP solved to its (bug-compatible) end: 214 796 pivots with thousands of closed
    batches and generic picks, many workgroups per launch. This is the case that exposed a state race
    in the blocked loop (a flag cleared by one workgroup while the others of the same launch still read
    it -- invisible in the 2000-pivot large-tableau check and in the small-LP checks): blocked
    and pipelined loops must end in the same status after the same number of pivots with the same
    tableau, and that is the oracle's."""
    import xpoly_amd
    leq, tg = gen.hard_lp_f64(300, 300)
    got = {}
    for mode in ("pipe", "block"):
        monkeypatch.setenv("XPG_LOOP", mode)
        c = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
        st = lp.two_stage()
        got[mode] = (st, lp.pivots_done(), lp.read())
        lp.close(); c.close()
    want = port.two_stage(F64, leq, tg, 0xFFFFFFFF)
    for mode, (st, piv, out) in got.items():
        assert st == want["status"] and piv == got["pipe"][1], (mode, st, piv)
        if want["status"] != 2:
            assert np.array_equal(out["tab"].view(np.uint64), want["tab"].view(np.uint64)), mode
            assert np.array_equal(out["tgtf"].view(np.uint64), want["tgtf"].view(np.uint64)), mode
            assert np.array_equal(out["eq2bv"], want["eq2bv"]), mode


@pytest.mark.parametrize("B", [1, 3, 9, 16, 24, 32])
def test_blocked_loop_small_and_rare_branches(ctx, port, B, monkeypatch):
    """The blocked loop forced onto small LPs (where it is not the default), batch lengths 1, 3, 9, 16, 24
    and 32 (the last three have full-batch passes of their own; batches that close with 17 .. 31 pivots staged
    take the run-time-length pass): dependence-test-like data drive it through closed batches and the generic pick; random
    problems through phase 1. Status, tableau, objective row, basis: bit-identical to the oracle."""
    import xpoly_amd
    monkeypatch.setenv("XPG_LOOP", "block")
    monkeypatch.setenv("XPG_BLOCK", str(B))
    c = xpoly_amd.Context(0)
    six = xpoly_amd.SIX(c, F64)
    leqs, tgs = gen.small_lp_batch_f64(4, 20, 29, family=1, seed=gen.XS_SEED + 91)
    rng = np.random.default_rng(5 + B)
    cases = [(leqs[b], tgs[b]) for b in range(4)]
    for _ in range(6):
        p = gen.random_problem(rng, F64, int(rng.integers(0, 3)), int(rng.integers(2, 14)), int(rng.integers(2, 14)), plain=True)
        cases.append((p["leq"], p["tgtf"]))
    for leq, tg in cases:
        for K in (5, 37, 0xFFFFFFFF):
            want = port.two_stage(F64, leq, tg, K)
            six.set_param(0, K)
            got = six.TwoStageMethod(leq, tg)
            assert got["status"] == want["status"], (B, K, got["status"], want["status"])
            if want["status"] == 2:
                continue
            for k in ("tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv"):
                a, b = np.asarray(got[k]), np.asarray(want[k])
                assert a.shape == b.shape and (np.array_equal(a.view(np.uint64), b.view(np.uint64))
                                               if a.dtype == np.float64 else np.array_equal(a, b)), (B, K, k)
    c.close()


@pytest.mark.parametrize("shape", [(511, 512), (1023, 1024), (640, 383), (300, 723)])
@needs_hooks
def test_chain_with_and_without_the_column_line_equals_the_pipelined_loop(shape, monkeypatch):
    """Round 5's chain -- replays from registers with -0.0 for the steps that do not count, one-round gathers, and where the row
    stride is a multiple of 4 KiB (the first three shapes: W = 1024 / 2048 / 1024) the entering column's line in the pick
    workers' LDS -- against the pipelined loop, which shares none of that code: whole solves of LPs that run thousands of
    pivots (rows that pivot several times inside one batch, columns that come back), state compared bit for bit at three
    iteration limits; with XPG_CHAIN_LINE forced off, on, and on as a half line."""
    import xpoly_amd
    m, n = shape
    leq, tg = gen.hard_lp_f64(m, n)
    monkeypatch.setenv("XPG_LOOP", "pipe")
    cp = xpoly_amd.Context(0)
    want = {}
    for K in (77, 1200, 3000):
        lp = xpoly_amd.DeviceLP(cp, F64, leq, tg)
        st = lp.two_stage(K)
        want[K] = (st, lp.read())
        lp.close()
    cp.close()
    monkeypatch.setenv("XPG_LOOP", "block")
    for line in ("0", "1", "8"):                         # off, the whole line, the half line (what 4096 x 8192 takes at 32 stages)
        monkeypatch.setenv("XPG_CHAIN_LINE", line)
        cb = xpoly_amd.Context(0)
        for K in (77, 1200, 3000):
            lp = xpoly_amd.DeviceLP(cb, F64, leq, tg)
            st = lp.two_stage(K)
            got = lp.read()
            lp.chain_aborts()
            runs = lp.chain_runs
            lp.close()
            assert st == want[K][0], (shape, line, K, st, want[K][0])
            for k in ("tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv"):
                a, b = np.asarray(got[k]), np.asarray(want[K][1][k])
                assert a.shape == b.shape and (np.array_equal(a.view(np.uint64), b.view(np.uint64))
                                               if a.dtype == np.float64 else np.array_equal(a, b)), (shape, line, K, k)
            if K > 100:
                assert runs > 0, (shape, line, K, runs)         # the chain launches really ran (not the launch-per-stage fallback)
        cb.close()


def test_batch_sizes_ragged_and_single(ctx, port):
    """nb = 1, nb not a multiple of anything, 1 x 1 LPs."""
    rng = np.random.default_rng(4)
    for nb, m, nv in ((1, 1, 1), (3, 2, 5), (130, 4, 3)):
        probs = [gen.random_problem(rng, RAT, 1, m, nv, plain=True) for _ in range(nb)]
        leq = np.stack([p["leq"] for p in probs]); tg = np.stack([p["tgtf"] for p in probs])
        for is_max in (True, False):
            status, v, sol = ctx.six_batch(RAT, is_max, tg, leq)
            for b in range(nb):
                want = port.six_solve(RAT, is_max, probs[b]["tgtf"], probs[b]["vc"], None, probs[b]["leq"])
                assert status[b] == want[0] and np.array_equal(v[b], want[1])


@needs_hooks
def test_wide_tableau_keeps_the_chain(port, monkeypatch):
    """W >= 16 384 (here 1024 x 17 025: 267 prep workers, 268 partial slots -- two polling rounds per pick) used to drop
    silently to the launch-per-stage kernels; the chain now runs there too, and its results are the serial loop's and
    the oracle's bit for bit."""
    import xpoly_amd
    m, n, K = 1024, 16000, 160
    leq, tg = gen.hard_lp_f64(m, n)
    got = {}
    for mode in ("block", "serial"):
        monkeypatch.setenv("XPG_LOOP", mode)
        c = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
        lp.begin()
        for k in (50, 110):
            assert lp.iterate(k) == xpoly_amd.six.XPG_RUNNING
        aborts, off = lp.chain_aborts()
        got[mode] = (lp.read(), lp.trace().copy(), lp.pivots_done(), lp.chain_runs, aborts)
        lp.close(); c.close()
    a, ta, na, runs, aborts = got["block"]
    b, tb, nb, _, _ = got["serial"]
    assert runs >= 5 and aborts == 0, (runs, aborts)     # the persistent launch really ran (2 x 24 + 2, then 4 x 24 + 14 pivots)
    assert na == nb == K and np.array_equal(ta, tb)
    for k in ("tab", "tgtf"):
        assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), k
    want = port.two_stage(F64, leq, tg, K)
    assert np.array_equal(a["tab"].view(np.uint64), want["tab"].view(np.uint64))
    assert np.array_equal(a["eq2bv"], want["eq2bv"])


def test_tableau_wider_than_the_partial_slots_of_one_wave_per_64_columns(port, monkeypatch):
    """Round 6 (found by bench.py's `shapes` leg): a tableau of W > 32 640 columns -- 96 x 33 097 here -- has more 64-column prep
    workgroups than the 510 look-ahead partial slots; the blocked loop ran over them (a GPU memory fault at 1024 x 33 793).
    Its prep workgroups now take 256 columns each there (launch-per-stage form; the chain's hand-off areas end at 32 640).
    Forced and automatic loop choice against the pipelined loop and the oracle, bit for bit. (The reference itself has no
    defined behaviour at this width: its vc matrix of (n + m)^2 cells overflows a 32-bit byte count from n + m = 23 171.)"""
    import xpoly_amd
    m, n, K = 96, 33000, 70
    leq, tg = gen.hard_lp_f64(m, n)
    want = port.two_stage(F64, leq, tg, K)
    for mode in ("block", "pipe", None):
        if mode is None:
            monkeypatch.delenv("XPG_LOOP", raising=False)
        else:
            monkeypatch.setenv("XPG_LOOP", mode)
        c = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
        lp.begin()
        info = lp.loop_info()
        if mode == "block":
            assert info["loop"] == "blocked" and info["chain"] == "launch per stage", info
        for k in (33, K - 33):
            assert lp.iterate(k) == xpoly_amd.six.XPG_RUNNING
        got = lp.read()
        assert lp.pivots_done() == K
        for k in ("tab", "tgtf"):
            assert np.array_equal(got[k].view(np.uint64), want[k].view(np.uint64)), (mode, k)
        assert np.array_equal(got["eq2bv"], want["eq2bv"]), mode
        lp.close(); c.close()


def test_fp64_loop_on_tableaux_with_inf_and_nan_cells(ctx, port):
    """A Float tableau may hold inf / NaN (convertEq2Ineq divides by an equality entry that can be 0, lpsol.h:1232).
    The reference then carries NaN through is_feasible's row sums (a nonbasic inf times an exact 0) and meets NaN
    ratios in findPivotBV's scan, where only its own row order decides: statuses at every iteration limit, tableau,
    objective row and basis must still be the oracle's (NaN = NaN: their signs and payloads differ between x86 and
    the GPU). Regression test of round 3: k_rowcheck skipped nonbasic terms whatever they held."""
    import xpoly_amd
    F64 = 0
    six = xpoly_amd.SIX(ctx, F64)
    rng = np.random.default_rng(3)

    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            return False
        if a.dtype != np.float64:
            return a.tobytes() == b.tobytes()
        na, nb = np.isnan(a), np.isnan(b)
        return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
    seen = set()
    for it in range(60):
        m, n = int(rng.integers(2, 8)), int(rng.integers(2, 7))
        A = rng.integers(-3, 6, size=(m, n)).astype(np.float64)
        A[rng.random((m, n)) < 0.3] = 0
        b = rng.integers(-2, 9, size=m).astype(np.float64)
        c = rng.integers(-2, 5, size=n).astype(np.float64)
        leq = np.concatenate([A, b[:, None]], axis=1)
        tg = np.concatenate([c, [0.0]])
        for _ in range(int(rng.integers(1, 3))):
            leq[int(rng.integers(0, m)), int(rng.integers(0, n + 1))] = rng.choice([np.inf, -np.inf, np.nan])
        for K in (0, 1, 2, 3, 5, 1000):
            want = port.two_stage(F64, leq, tg, K)
            if want["status"] == -7:
                continue
            six.set_param(0, K)
            got = six.TwoStageMethod(leq, tg)
            assert got["status"] == want["status"], (it, K, got["status"], want["status"])
            seen.add(want["status"])
            if want["status"] != 2:
                for k in ("tab", "tgtf", "eq2bv"):
                    assert same(got[k], want[k]), (it, K, k)
    assert len(seen) >= 3


@pytest.mark.parametrize("chain", ["1", "0"])
def test_blocked_loop_meets_nan_that_arises_mid_solve(port, monkeypatch, chain):
    """FINITE input that overflows after a few pivots (gen.overflow_lp_f64: magnitudes spread over hundreds of decades):
    inf - inf leaves NaN cells, findPivotBV meets NaN ratios, and its answer then depends on the order of its own scan
    (`minbval > v` is false either way round, lpsol.h:599-611; flty.cpp:61-94). The blocked loop chooses rows by
    reductions over records, which cannot say "unordered": a pick that meets a NaN candidate ends the batch and hands the
    pivot to the generic pick, which scans in order (lp_chain.hip.h ch_poll_records, lp_blocked.hip.h blk_pick_body).
    States at iteration limits around the first NaN and at the end must be the oracle's -- which equals the real reference
    on these very LPs (tools/crosscheck_oracle_weird.py overflow: 21 LPs x 6 limits, 0 mismatches). With the persistent
    chain launch and with the launch-per-stage kernels (XPG_CHAIN=0)."""
    import xpoly_amd
    monkeypatch.setenv("XPG_LOOP", "block")              # read when the context is created
    monkeypatch.setenv("XPG_CHAIN", chain)
    c = xpoly_amd.Context(0)
    six = xpoly_amd.SIX(c, F64)

    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            return False
        if a.dtype != np.float64:
            return a.tobytes() == b.tobytes()
        na, nb = np.isnan(a), np.isnan(b)                # NaN signs and payloads differ between x86 and the GPU
        return np.array_equal(na, nb) and a[~na].tobytes() == b[~nb].tobytes()
    cases = compared = 0
    for trial in range(1200):
        leq, tg = gen.overflow_lp_f64(trial)
        assert np.isfinite(leq).all() and np.isfinite(tg).all()
        if not np.isnan(port.two_stage(F64, leq, tg, 0xFFFFFFFF)["tab"]).any():
            continue
        lo = next(K for K in range(1, 1000) if np.isnan(port.two_stage(F64, leq, tg, K)["tab"]).any())
        cases += 1
        for K in sorted({max(1, lo - 1), lo, lo + 1, lo + 2, lo + 5, 0xFFFFFFFF}):
            want = port.two_stage(F64, leq, tg, K)
            six.set_param(0, K)
            got = six.TwoStageMethod(leq, tg)
            assert got["status"] == want["status"], (trial, K, lo, got["status"], want["status"])
            if want["status"] != 2:
                for k in ("tab", "tgtf", "eq2bv"):
                    assert same(got[k], want[k]), (trial, K, lo, k)
            compared += 1
    c.close()
    assert cases >= 20 and compared >= 6 * 20 - 20
