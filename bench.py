#!/usr/bin/env python3
"""bench.py -- simplex pivots/sec on a fp64 4096 x 8192 tableau (BASELINE.json's metric),
plus batched small-LP throughput, on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE simplex pivot of the device-resident loop -- pricing, ratio test,
pivot-pair upkeep, row/column staging and the rank-1 tableau update
(src/com/lpsol.h:1039-1188) -- on a dense LP with m = 4096 constraints and
n = 4095 variables, whose slack tableau is exactly 4096 x 8192 fp64 (268 MB,
resident in HBM before the timed region). A single tableau does not shard
("replicas only"): with N ranks every rank runs its own replica and
`value` is the sum. The batched leg (BASELINE.json configs[2]: independent 32 x 64
LPs, the dependence-test shape, 8192 per GPU = 65 536 on 8 GPUs) shards contiguously
(xpoly_amd/shard.py) with no data-path collective and ONE all_gather of the result
records at the end, inside its timed region; it is reported under "batched".

One JSON line is printed by rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, NVARS = 4096, 4095                      # tableau 4096 x (4095 + 4096 + 1) = 4096 x 8192
TAB_W = NVARS + M + 1
ALG_BYTES_PER_PIVOT = 2 * M * TAB_W * 8    # every entry read once and written once
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
BATCH_PER_GPU = 8192                       # 65 536 LPs over 8 GPUs (BASELINE.json configs[2])
BATCH_M, BATCH_COLS = 32, 64
PREWARM = 256                              # untimed set-up iterations before the warmup
PREWARM_SECONDS = 0.5
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round1_pmc_hbm_traffic.json")


def cpu_baseline_pivots(budget_s=12.0):
    """The oracle's K1 (oracle/oracle.cpp orc_pivot_f64) on a 4096 x 8192 tableau, 1 core."""
    from oracle.checker import Port
    from tools import gen
    port = Port()
    tab, obj = gen.tableau_f64(M, TAB_W)
    fn = port.lib.orc_pivot_f64
    n, t0 = 0, time.perf_counter()
    rng = np.random.default_rng(0)
    while True:
        r, c = int(rng.integers(0, M)), int(rng.integers(0, TAB_W - 1))
        fn(tab.ctypes.data_as(C.c_void_p), C.c_int(M), C.c_int(TAB_W), obj.ctypes.data_as(C.c_void_p),
           C.c_int(TAB_W - 1), C.c_int(r), C.c_int(c))
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s and n >= 3:
            break
    return dict(value=round(n / dt, 3), unit="pivots/s", cores=1, kind="port",
                sample="%d pivots of the oracle's SIX::pivot restatement (orc_pivot_f64) on a 4096x8192 fp64 "
                       "tableau, 1 thread, %.1f s" % (n, dt))


def cpu_reference_pivots(leq, tgtf):
    """The REAL reference (oracle/_ref/libxpoly_ref.so, if it travelled): SIX::TwoStageMethod with
    set_param(0, K) for K = 1 and K = 3 on the same LP; per-pivot time by differencing."""
    from oracle.checker import F64, Ref
    if not Ref.available():
        return None
    ref = Ref()
    ts = {}
    for K in (1, 3):
        t0 = time.perf_counter()
        ref.two_stage(F64, leq, tgtf, K)
        ts[K] = time.perf_counter() - t0
    per = (ts[3] - ts[1]) / 2.0
    return dict(value=round(1.0 / per, 4) if per > 0 else None, unit="pivots/s", cores=1, kind="reference",
                sample="xcom::SIX<FloatMat,Float>::TwoStageMethod on the bench LP (m=4096,n=4095), "
                       "(t[K=3]-t[K=1])/2 = %.3f s/pivot; set-up %.1f s per call" % (per, ts[1] - per))


def cpu_baseline_batch(leq, tgtf, budget_s=6.0):
    from oracle.checker import Port
    from tools import gen
    port = Port()
    vc = gen.vc_nonneg(BATCH_COLS - 1)
    n, t0 = 0, time.perf_counter()
    while n < len(leq):
        port.six_solve(0, True, tgtf[n], vc, None, leq[n])
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 8:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 2), unit="LPs/s", cores=1, kind="port",
                sample="%d LPs (32x64, SIX::maxm) through the oracle, 1 thread, %.1f s" % (n, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-baseline", action="store_true",
                    help="skip timing the real reference build (oracle/_ref, ~1 s on the GPU box's host)")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="do not bracket sweep launches with HIP events")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    import xpoly_amd
    from tools import gen
    from xpoly_amd.shard import gather_records, pack_records, shard_range
    RUNNING = xpoly_amd.six.XPG_RUNNING
    ctx = xpoly_amd.Context(dev.index)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- leg 1: pivots/s on the 4096 x 8192 tableau (one replica per rank) -------------------
    leq, tgtf = gen.hard_lp_f64(M, NVARS)      # the same LP on every rank (identical replicas)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tgtf)
    lp.begin()
    state = dict(since_begin=0, restarts=0)
    LP_LIFE = 3800      # this LP reaches its (bug-compatible) end after 4165 pivots (tools/probe_count.py)

    def run_pivots(n):
        """Exactly n pivots of the device loop. Before the LP would reach its end the slack tableau
        is rebuilt on the device from the resident input (one 90 us kernel, inside the timed
        region when it happens there) and the loop continues, so every queued launch does work."""
        left = n
        while left > 0:
            if state["since_begin"] >= LP_LIFE:
                lp.begin()
                state["since_begin"] = 0
                state["restarts"] += 1
            chunk = min(left, LP_LIFE - state["since_begin"])
            st = lp.iterate(chunk)
            assert st == RUNNING, "LP ended early (status %d)" % st
            state["since_begin"] += chunk
            left -= chunk

    # set-up, not measured: one pass through every code path of the timed region (launch
    # throttling, event pairs) so lazy runtime initialisation does not land inside it
    ctx.profile_begin(8, 32)
    # (the ROCm runtime was measured to stall the stream once for 60-80 ms somewhere in the first
    # ~100 ms of queued-loop activity of a process -- tools/probe_stall.py -- so the set-up pass runs
    # for at least PREWARM_SECONDS of wall time, not just PREWARM iterations)
    t_pre = time.perf_counter()
    run_pivots(PREWARM)
    ctx.sync()
    while time.perf_counter() - t_pre < PREWARM_SECONDS:
        run_pivots(PREWARM)
        ctx.sync()
    ctx.profile_end()
    run_pivots(a.warmup)
    barrier()
    stride = max(1, a.steps // 320)           # sampled sweep launches spread over the region (a blocked
                                              # sweep applies up to 16 pivots, so there are ~steps/16 of them:
                                              # about 20 event pairs per 1000 steps)
    ctx.profile_begin(0 if a.no_events else a.steps, stride)
    start_count = lp.pivots_done()
    t0 = time.perf_counter()
    run_pivots(a.steps)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    launches, sweep_ms = ctx.profile_end()
    done = lp.pivots_done() - start_count
    assert done == a.steps, "expected %d pivots, device did %d" % (a.steps, done)
    rows, W, rhs = lp.shape()
    assert (rows, W) == (M, TAB_W)
    dt = max_over_ranks(dt)
    value = world * a.steps / dt
    roofline = None
    if launches:
        sweep_avg_s = sweep_ms / 1e3 / launches
        achieved = ALG_BYTES_PER_PIVOT / sweep_avg_s / 1e9
        traffic, traffic_src = None, None
        if os.path.exists(PMC_SUMMARY):      # measured with rocprofv3 --pmc (separate passes), not live
            pm = json.load(open(PMC_SUMMARY))
            traffic = round(pm["traffic_bytes_per_launch"])
            traffic_src = "profiles/round1_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, KiB->B)"
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                        kernel={"se": "k_update_f64<32,8>", "pi": "k_pipe_sweep<32,8> (sweep + the next pivot's pick workgroups)",
                                "sp": "k_pipe_sweep<32,8>"}.get(os.environ.get("XPG_LOOP", "")[:2],
                                                               "k_blk_sweep_full<16,4> (one pass applies the XPG_BLOCK=16 staged pivots)"),
                        pivots_per_launch=round(a.steps / max(1, launches * stride), 2),
                        launches_sampled=launches,
                        avg_launch_us=round(sweep_avg_s * 1e6, 2),
                        algorithmic_bytes_per_launch=ALG_BYTES_PER_PIVOT)
    lp.close()

    # ---- leg 2: batched 32 x 64 LPs, sharded across ranks, one all_gather at the end ----------
    batched = None
    b_leq = b_tg = None
    if not a.no_batched:
        total = BATCH_PER_GPU * world
        lo, hi = shard_range(total, rank, world)
        nloc = hi - lo
        fams = {}
        for fam, name in ((0, "dense_positive"), (1, "dep_test_like")):
            b_leq, b_tg = gen.small_lp_batch_f64(nloc, BATCH_M, BATCH_COLS, fam,
                                                 seed=gen.XS_SEED + 1000 * (rank + 1) + fam)
            d_leq = torch.from_numpy(b_leq).to(dev)
            d_tg = torch.from_numpy(b_tg).to(dev)
            d_st = torch.empty(nloc, dtype=torch.int32, device=dev)
            d_v = torch.empty(nloc, dtype=torch.float64, device=dev)
            d_sol = torch.zeros(nloc, BATCH_COLS, dtype=torch.float64, device=dev)
            d_piv = torch.empty(nloc, dtype=torch.int32, device=dev)
            full = None

            def one_pass():
                ctx.six_batch_dev(xpoly_amd.F64, True, nloc, d_tg.data_ptr(), d_leq.data_ptr(),
                                  BATCH_M, BATCH_COLS, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(),
                                  d_piv.data_ptr())
                ctx.sync()
                if dist is None:
                    return None
                # the only collective of the path: fixed-size (status, v, sol) records
                out = gather_records(pack_records(d_st, d_v, d_sol), total, rank, world, dist)
                torch.cuda.synchronize()
                return out

            full = one_pass()
            barrier()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                full = one_pass()
            barrier()
            bdt = max_over_ranks(time.perf_counter() - t0)
            if full is not None:
                assert full.shape[0] == total
            piv = float(d_piv.sum().item())
            if dist is not None:
                pt = torch.tensor([piv], dtype=torch.float64, device=dev)
                dist.all_reduce(pt)
                piv = float(pt.item())
            hist = torch.bincount(d_st.clamp(min=0), minlength=5).tolist()
            fams[name] = dict(lps_per_s=round(total * reps / bdt, 1), pivots_per_s=round(piv * reps / bdt, 1),
                              status_hist_rank0=hist, ms_per_pass=round(bdt / reps * 1e3, 3))
        tot = sum(f["lps_per_s"] for f in fams.values()) / len(fams)
        batched = dict(metric="batched LPs/sec", value=round(tot, 1), unit="LPs/s", n_gpus=world,
                       total_lps=total, lps_per_gpu=BATCH_PER_GPU,
                       shape="leq 32x64 (63 vars + rhs), SIX::maxm, x>=0, whole solve per LP in LDS",
                       scaling="weak",
                       collective="one all_gather_into_tensor of (status,v,sol) records" if dist else "none (1 GPU)",
                       families=fams)

    # ---- CPU baseline (rank 0, N = 1 only) -----------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline_pivots()
        if not a.no_ref_baseline:
            cpu["reference"] = cpu_reference_pivots(leq, tgtf)
        if b_leq is not None:
            cpu["batched"] = cpu_baseline_batch(b_leq, b_tg)

    if rank == 0:
        out = {
            "metric": "simplex pivots/sec (float tableau 4kx8k)",
            "value": round(value, 2), "unit": "pivots/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "dense LP m=4096 n=4095 (gen.hard_lp_f64: A~U(0.1,1), b=A x*, c=A^T y*), "
                                   "slack tableau 4096x8192 fp64 resident in HBM, device-resident "
                                   "SIX::solveSlackForm loop, one pivot per step",
                       "tableau": [M, TAB_W], "parallelism": "replicas only (1 tableau per GPU)",
                       "lp_restarts_in_run": state["restarts"]},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "batched": batched,
        }
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
