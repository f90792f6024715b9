"""The one-launch-per-pivot Rational loop (csrc/lp_fused_r32.hip.h) beyond the 1024 x 2048 fixture: mid-size LPs whose
launches hold several pick waves and stager workgroups, 7 / 12 / 90 pivots (odd and even counts: either side of the
ping-pong tableau is the one read back; phase one with ties and degenerate steps in two of the families) -- against the
oracle, and against the two-launch loop and other spacings of the generic point, which must give the same pivots. (Whole
solves with every deferred decision -- relaxed ratio pass, disableNV, findPivotNVandBVPair -- are the small LPs of
test_gpu_parity.py::test_two_stage_matches_oracle, which run on this loop too.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tools import gen

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RAT = 1


def run_worker(**env_over):
    from conftest import hooks_env
    env = hooks_env()                                    # (the forced slices / loop forms are hook-only switches: the -DXPG_TEST_HOOKS build)
    for k in ("XPG_R32_LOOP", "XPG_R32_GENERIC_EVERY"):
        env.pop(k, None)
    env.update(env_over)
    r = subprocess.run([sys.executable, os.path.join(HERE, "fused_rational_worker.py")], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_fused_loop_matches_oracle_and_the_other_loops(port):
    import zlib
    sys.path.insert(0, HERE)
    import fused_rational_worker as W

    def crc(a):
        return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)

    fused = run_worker(XPG_R32_LOOP="fused")             # (forced: the size rule sends LPs this small to the two-launch loop)
    assert len(fused) == len(W.CASES) * len(W.KS)
    it = iter(fused)
    for fam, m, nv, prob in W.problems():
        for K in W.KS:
            rec = next(it)
            want = port.two_stage(RAT, prob["leq"], prob["tgtf"], K)
            assert rec["status"] == want["status"], (fam, m, nv, K, rec["status"], want["status"])
            if want["status"] != 2:
                assert rec["tab"] == crc(want["tab"]), (fam, m, nv, K)
                assert rec["tgtf"] == crc(want["tgtf"]), (fam, m, nv, K)
                assert rec["eq2bv"] == crc(np.asarray(want["eq2bv"], dtype=np.int32)), (fam, m, nv, K)
    # the same pivots from the two-launch loop and with a generic point before every / every third launch
    for env in (dict(XPG_R32_LOOP="pipe"), dict(XPG_R32_LOOP="fused", XPG_R32_GENERIC_EVERY="1"), dict(XPG_R32_LOOP="fused", XPG_R32_GENERIC_EVERY="3"), {}):
        other = run_worker(**env)
        assert len(other) == len(fused)
        for a, b in zip(fused, other):
            assert a == b, (env, a["fam"], a["m"], a["nv"], a["K"])
    assert any(len(r["trace"]) >= 2 * 90 for r in fused)
    assert any("tab" in r for r in fused)


def test_fused_loop_small_whole_solves_match_oracle(port):
    """Whole solves of the small random LPs of test_gpu_parity.py (stage 1, every deferred decision) forced through the
    fused loop."""
    import zlib
    sys.path.insert(0, HERE)
    import fused_rational_worker as W

    def crc(a):
        return "%08x" % (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)

    got = run_worker(XPG_R32_LOOP="fused", XPG_FUSED_SMALL="1")
    it = iter(got)
    n = 0
    for fam, k, prob in W.small_problems():
        for K in W.SMALL_KS:
            rec = next(it)
            want = port.two_stage(RAT, prob["leq"], prob["tgtf"], K)
            assert rec["status"] == want["status"], (fam, k, K, rec["status"], want["status"])
            if want["status"] != 2:
                assert rec["tab"] == crc(want["tab"]) and rec["tgtf"] == crc(want["tgtf"]), (fam, k, K)
                assert rec["eq2bv"] == crc(np.asarray(want["eq2bv"], dtype=np.int32)), (fam, k, K)
            n += 1
    assert n == len(got) == 3 * 12 * len(W.SMALL_KS)


def test_rational_iterate_in_chunks_equals_one_shot(ctx):
    """Chunks of 1, 2, 3, ... launches (every call starts with a generic point and ends on either side of the ping-pong
    tableau) leave the state one call of the same length leaves."""
    import xpoly_amd
    leq, tg = gen.int_lp_rat(512, 700)                   # 512 x 1213: a size the fused loop takes by default
    a = xpoly_amd.DeviceLP(ctx, RAT, leq, tg); a.begin()
    for k in (1, 2, 3, 1, 5, 4, 1):
        assert a.iterate(k) == xpoly_amd.six.XPG_RUNNING
        mid = a.read()                                   # a read between the calls must not disturb the loop
        assert mid["tab"].shape[0] == 512
    b = xpoly_amd.DeviceLP(ctx, RAT, leq, tg); b.begin()
    assert b.iterate(17) == xpoly_amd.six.XPG_RUNNING
    ra, rb = a.read(), b.read()
    assert a.pivots_done() == b.pivots_done() == 17
    assert np.array_equal(a.trace(), b.trace())
    for k in ("tab", "tgtf", "nvset", "bvset", "bv2eq", "eq2bv"):
        assert np.array_equal(ra[k], rb[k]), k
    a.close(); b.close()
