"""GPU: what round 2 added around the kernels -- the multi-device C-ABI batches (one context + host
thread per shard inside the library), the per-handle device binding of every entry point, and the
blocked loop's sweep counters."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import needs_hooks

from tools import gen

pytestmark = pytest.mark.gpu

F64, RAT = 0, 1


def bits(a):
    a = np.asarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def _devices():
    """Two shards even on a 1-GPU box (a device may be listed twice); real devices 0, 1 when there are two."""
    from xpoly_amd._capi import lib
    return [0, 1] if lib().xpg_device_count() >= 2 else [0, 0]


@pytest.mark.parametrize("kind", [F64, RAT])
@pytest.mark.parametrize("nb", [1, 2, 37])
def test_six_batch_multi_equals_single(ctx, kind, nb):
    """xpg_six_batch_*_multi: contiguous shards, results written straight into the caller's arrays in
    the global order -- identical to one xpg_six_batch_* call, ragged and tiny batches included."""
    from xpoly_amd.six import six_batch_multi
    if kind == F64:
        leq, tg = gen.small_lp_batch_f64(nb, 12, 17, 1, seed=gen.XS_SEED + nb)
    else:
        rng = np.random.default_rng(nb)
        probs = [gen.random_problem(rng, RAT, 1, 6, 5, plain=True) for _ in range(nb)]
        leq = np.stack([p["leq"] for p in probs]); tg = np.stack([p["tgtf"] for p in probs])
    for is_max in (True, False):
        want = ctx.six_batch(kind, is_max, tg, leq)
        for devs in (_devices(), [0], [0, 0, 0]):
            got = six_batch_multi(devs, kind, is_max, tg, leq)
            assert np.array_equal(got[0], want[0]), (devs, is_max)
            assert np.array_equal(bits(got[1]), bits(want[1]))
            ok = want[0] == 0
            assert np.array_equal(bits(got[2])[ok], bits(want[2])[ok])


def test_mip_and_dep_batch_multi_equal_single(ctx):
    from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_batch_multi, mip_batch, mip_batch_multi
    leq, tg = gen.knapsack_batch_rat(21, 6, seed=gen.XS_SEED + 3)
    want = mip_batch(ctx, True, True, tg, leq)
    got = mip_batch_multi(_devices(), True, True, tg, leq)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[3] == want[3]
    assert np.array_equal(got[2][want[0] == 0], want[2][want[0] == 0])
    rng = np.random.default_rng(8)
    mats = np.stack([gen.random_system(rng, 8, 3) for _ in range(33)])
    mats[..., 1] = 1
    w_empty, w_nodes = dep_is_empty_batch(ctx, mats)
    g_empty, g_nodes = dep_is_empty_batch_multi(_devices(), mats)
    assert np.array_equal(g_empty, w_empty) and g_nodes == w_nodes


def test_eight_shards_of_a_whole_node_batch_equal_the_single_call(ctx):
    """The node-level shapes without an 8-GPU node: `ndev = 8` (device 0 eight times -- eight contexts, eight host threads,
    eight concurrent launches on one GPU) on batches that do not divide by 8: 65 537 LPs (configs[2]'s batch + 1: shards of
    8193 and 8192), 1 025 0-1 knapsack trees, 4 097 dependence polyhedra -- every status, optimum, solution, verdict and node
    count identical to the single-device call, in the caller's order."""
    from xpoly_amd.six import dep_is_empty_batch, dep_is_empty_batch_multi, mip_batch, mip_batch_multi, six_batch_multi
    eight = [0] * 8
    nb = 65537
    leq, tg = gen.small_lp_batch_f64(nb, 12, 17, 1, seed=gen.XS_SEED + 4242)
    want = ctx.six_batch(F64, True, tg, leq)
    got = six_batch_multi(eight, F64, True, tg, leq)
    assert np.array_equal(got[0], want[0]) and np.array_equal(bits(got[1]), bits(want[1]))
    ok = want[0] == 0
    assert ok.sum() > 100, np.bincount(want[0].clip(0), minlength=5)
    assert np.array_equal(bits(got[2])[ok], bits(want[2])[ok])
    del leq, tg, want, got
    leq, tg = gen.knapsack_batch_rat(1025, 16, seed=gen.XS_SEED + 5)
    want = mip_batch(ctx, True, True, tg, leq)
    got = mip_batch_multi(eight, True, True, tg, leq)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[3] == want[3]
    assert np.array_equal(got[2][want[0] == 0], want[2][want[0] == 0])
    rng = np.random.default_rng(88)
    mats = np.stack([gen.random_system(rng, 10, 4) for _ in range(4097)])
    mats[..., 1] = 1
    w_empty, w_nodes = dep_is_empty_batch(ctx, mats)
    g_empty, g_nodes = dep_is_empty_batch_multi(eight, mats)
    assert np.array_equal(g_empty, w_empty) and g_nodes == w_nodes


def test_multi_reports_a_missing_device():
    from xpoly_amd import XpgError
    from xpoly_amd._capi import lib
    from xpoly_amd.six import six_batch_multi
    leq, tg = gen.small_lp_batch_f64(4, 4, 5, 0)
    with pytest.raises(XpgError, match="XPG_ERR_NO_DEVICE"):
        six_batch_multi([0, lib().xpg_device_count()], F64, True, tg, leq)


def test_entry_points_bind_their_own_device_and_restore_the_callers(ctx, port):
    """Every extern "C" entry point runs on its handle's device whatever the calling thread had selected
    and leaves that selection as it found it. With two GPUs: a handle on device 1 is
    driven while the thread's current device is 0; with one GPU only the restore can be observed."""
    import xpoly_amd
    from xpoly_amd._capi import lib
    hip = C.CDLL("libamdhip64.so")                       # the runtime the library itself is linked against

    def current_device():
        d = C.c_int(-1)
        assert hip.hipGetDevice(C.byref(d)) == 0
        return d.value

    ndev = lib().xpg_device_count()
    target = 1 if ndev >= 2 else 0
    assert hip.hipSetDevice(0) == 0
    c = xpoly_amd.Context(target)
    assert current_device() == 0
    leq, tg = gen.dense_lp_f64(24, 40)
    got = xpoly_amd.SIX(c, F64).TwoStageMethod(leq, tg)
    want = port.two_stage(F64, leq, tg, 0xFFFFFFFF)
    assert got["status"] == want["status"]
    assert np.array_equal(bits(got["tab"]), bits(want["tab"]))
    bl, bt = gen.small_lp_batch_f64(16, 8, 12, 0)
    st, v, sol = c.six_batch(F64, True, bt, bl)          # hipMalloc + dynamic-LDS attribute + launch
    w = ctx.six_batch(F64, True, bt, bl)
    assert np.array_equal(st, w[0]) and np.array_equal(bits(v), bits(w[1]))
    tab, obj = gen.tableau_f64(40, 64)
    t2, o2 = c.pivot(F64, tab.copy(), obj.copy(), 63, 3, 5)
    t1, o1 = ctx.pivot(F64, tab.copy(), obj.copy(), 63, 3, 5)
    assert np.array_equal(bits(t1), bits(t2)) and np.array_equal(bits(o1), bits(o2))
    assert current_device() == 0
    c.close()
    assert current_device() == 0


def test_sweep_counters_and_tail_batches(monkeypatch):
    """xpg_lp_counters: 100 iterations of the blocked loop are 3 full sweeps (32 pivots each, the default) and one of 4
    pivots (the tail of the budget is enqueued at its own length); 64 more are two full ones and no tail. The
    same with batches of 16 (XPG_BLOCK), where 64 is a whole number of batches. The tableau is the pipelined loop's
    either way."""
    import xpoly_amd
    leq, tg = gen.hard_lp_f64(96, 120)
    out = {}
    for mode in ("block", "block16", "pipe"):
        monkeypatch.setenv("XPG_LOOP", mode[:5])
        if mode == "block16":
            monkeypatch.setenv("XPG_BLOCK", "16")
        else:
            monkeypatch.delenv("XPG_BLOCK", raising=False)
        c = xpoly_amd.Context(0)
        lp = xpoly_amd.DeviceLP(c, F64, leq, tg)
        lp.begin()
        assert lp.iterate(100) == xpoly_amd.six.XPG_RUNNING
        if mode != "pipe":
            assert lp.counters() == ((3, 1) if mode == "block" else (6, 1))
        assert lp.iterate(64) == xpoly_amd.six.XPG_RUNNING
        if mode != "pipe":
            assert lp.counters() == ((5, 1) if mode == "block" else (10, 1))
        out[mode] = lp.read()
        assert lp.pivots_done() == 164
        lp.close(); c.close()
    for mode in ("block", "block16"):
        assert np.array_equal(bits(out[mode]["tab"]), bits(out["pipe"]["tab"]))
        assert np.array_equal(bits(out[mode]["tgtf"]), bits(out["pipe"]["tgtf"]))


def test_bench_two_ranks_real_solver_on_one_gpu():
    """bench.py's whole N > 1 path with the REAL solver: two ranks started by bench.py itself, both on cuda:0
    (--same-device: the collectives then run over gloo on host tensors, RCCL refuses two ranks per GPU): contiguous
    shards, weak and strong batched legs, the exact int32-record legs, one all_gather each -- and bench.py's own
    self-check of the gathered records against the reference fixture."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
                        "--legs", "batched,sharded", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and "stub_solver" not in out
    b = out["batched"]
    assert b["ranks"] == 2 and b["total_lps"] == 16384 and b["families"]["dep_test_like"]["gather_ms"] > 0
    assert b["strong_scaling"]["lps_per_rank"] == 32768
    assert out["sharded"]["mip"]["problems_total"] == 2048 and out["sharded"]["dep_is_empty"]["problems_total"] == 8192
    assert "identical" in out["self_check"]["batched"]["dep_test_like"]


def test_bench_rccl_at_world_size_one():
    """RCCL itself, once, before the first 8-GPU run: bench.py --gpus 1 --force-dist spawns its rank through
    torch.distributed.run before any GPU call and then takes the N > 1 path with the real solver --
    init_process_group("nccl", device_id=cuda:0), record tensors resident on the device, all_reduce (timing) and
    all_gather_into_tensor (result records) over RCCL at world size 1 -- and bench.py's self-check reads the gathered
    records against the reference fixture. (No scaling can be measured on one GPU; this makes sure the 8-GPU driver
    run is not also the collective path's first run.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
                        "--legs", "batched,sharded", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "stub_solver" not in out
    b = out["batched"]
    assert b["ranks"] == 1 and "RCCL, world size 1" in b["collective"], b.get("collective")
    assert b["families"]["dep_test_like"]["gather_ms"] > 0
    assert "RCCL, world size 1" in out["sharded"]["collective"]
    assert "identical" in out["self_check"]["batched"]["dep_test_like"]
