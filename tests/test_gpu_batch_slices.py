"""k_batch's time slices (csrc/batch_kernels.hip.h): an LP that has run `slice` iterations of a solve gives its LDS slot
back (LDS block -> HBM, index -> queue) and continuation workgroups take queued LPs in turn. Results must not depend on
it: the same batches with slices off, with the default slice (only launches that hold more LPs than the chip seats), and
with forced tiny slices (5 and 37 iterations: every LP is handed back many times, in stage 1's solve and in its own, fp64
and Rational, primal and dual) give the same status / value / solution arrays bit for bit; the first LPs also against the
oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_worker(**env_over):
    from conftest import hooks_env
    env = hooks_env()                                    # (the forced slices / loop forms are hook-only switches: the -DXPG_TEST_HOOKS build)
    for k in ("XPG_BATCH_SLICE", "XPG_BATCH_SLICE_FORCE"):
        env.pop(k, None)
    env.update(env_over)
    r = subprocess.run([sys.executable, os.path.join(HERE, "batch_slices_worker.py")], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_time_slices_do_not_change_results(port):
    sys.path.insert(0, HERE)
    import batch_slices_worker as W
    off = run_worker(XPG_BATCH_SLICE="0")
    assert len(off) == len(W.CASES)
    dflt = run_worker()                                  # 1536-3072 LPs per launch: more than the chip seats, so sliced
    for a, b in zip(off, dflt):
        assert a == b, ("default slice", a["kind"], a["fam"], a["nb"], a["is_max"], a["hist"], b["hist"])
    small = run_worker(XPG_BATCH_SLICE="0", XPG_SLICE_TEST_SMALL="1")
    for env in (dict(XPG_BATCH_SLICE="5"), dict(XPG_BATCH_SLICE="37")):
        got = run_worker(XPG_BATCH_SLICE_FORCE="1", XPG_SLICE_TEST_SMALL="1", **env)
        assert len(got) == len(small)
        for a, b in zip(small, got):
            assert a == b, (env, a["kind"], a["fam"], a["nb"], a["is_max"], a["hist"], b["hist"])
    # anchor: the first LPs of every case against the oracle
    from tools import gen
    for rec, (kind, fam, nb, is_max, limit, leq, tg) in zip(off, W.problems()):
        vc = gen.vc_nonneg(63, kind_float=(kind == 0))
        if kind != 0:
            vc = gen.to_rat(vc)
        for b in range(W.HEAD):
            st, v, _ = port.six_solve(kind, is_max, tg[b], vc, None, leq[b], max_iter=limit)
            assert rec["head_status"][b] == st, (kind, fam, is_max, b)
            if st == 0:
                assert np.asarray(rec["head_v"][b]).tolist() == np.asarray(v).tolist(), (kind, fam, is_max, b)
    assert any(r["hist"][0] > 0 for r in off) and any(r["hist"][2] > 0 for r in off) and any(r["hist"][4] > 0 for r in off)
