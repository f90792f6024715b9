#!/usr/bin/env python3
"""bench.py -- simplex pivots/sec on a fp64 4096 x 8192 tableau (BASELINE.json's metric), batched
small-LP throughput sharded over the GPUs of one node, and one leg per remaining BASELINE config.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment bench.py starts its own N ranks (torch.distributed.run on
127.0.0.1, one rank per GPU, backend nccl = RCCL) BEFORE anything touches a GPU and relays rank 0's
JSON line; launched under torch.distributed.run by somebody else it just is a rank.

Leg 1 (the headline, `value`): a STEP is one run of the device-resident SIX::solveSlackForm loop
(src/com/lpsol.h:1039-1188: pricing, ratio test, pivot-pair upkeep, row / column staging, rank-1
tableau update) over PIVOTS_PER_STEP = 3840 pivots = 120 full batches of 32 on a dense LP with
m = 4096, n = 4095, whose slack tableau is exactly 4096 x 8192 fp64 (268 MB, resident in HBM before
the timed region): xpg_lp_begin rebuilds the slack form on the device from the resident input
(one kernel, inside the step) and xpg_lp_iterate(3840) runs the loop; the LP's bug-compatible end
comes after 4165 pivots, so a step never meets it. A single tableau does not shard ("replicas
only"): every rank runs its own replica and `value` is the sum.

Leg 2 (BASELINE configs[2]): independent 32 x 64 LPs, 8192 per GPU (65 536 on 8), contiguous shards
(xpoly_amd/shard.py), no data-path collective, ONE all_gather of the result records inside the
timed region. Legs 3-5 run at N = 1 only (or shard like leg 2 where the problems are independent):
cfg 2b (LP m=4096, n=8192: tableau 4096 x 12289), cfg 4 (exact rational simplex, tableau
1024 x 2048, K = 16), cfg 5 (0-1 MIP branch and bound, node LPs as GPU batches).

One JSON line is printed by rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, NVARS = 4096, 4095                      # tableau 4096 x (4095 + 4096 + 1) = 4096 x 8192
TAB_W = NVARS + M + 1
BLOCK = int(os.environ.get("XPG_BLOCK", "32"))   # pivots one blocked sweep applies (XPG_BLOCK default: 32)
PIVOTS_PER_STEP = 3840                     # 120 full batches of 32; the LP ends after 4165 (tools/lab/probe_count.py)
ALG_BYTES_PER_LAUNCH = 2 * M * TAB_W * 8   # one sweep LAUNCH reads and writes every entry once (SURVEY 8d:
                                           # 2*m*W*8 B; the blocked loop pays it per BLOCK pivots, not per pivot)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
PCIE_GBS = 63.0                            # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec)
BATCH_PER_GPU = 8192                       # 65 536 LPs over 8 GPUs (BASELINE.json configs[2])
BATCH_M, BATCH_COLS = 32, 64
PREWARM_SECONDS = 0.5
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round6_pmc_hbm_traffic_4096x8192.json")
PMC_SUMMARY_2B = os.path.join(ROOT, "profiles", "round6_pmc_hbm_traffic_4096x12289.json")
PMC_BATCH = os.path.join(ROOT, "profiles", "round5_pmc_batch_issue.json")
PMC_RATIONAL = os.path.join(ROOT, "profiles", "round4_pmc_rational_issue.json")
LEGS = ("pivots", "batched", "sharded", "cfg2b", "shapes", "six_e2e", "one_call", "rational", "mip", "lineq")
LINEQ_NB = 16384                           # systems per row-elimination launch (the dependence tests' shapes)


# ---------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the oracle (kind "port") and, when it travelled, the real
# reference build (kind "reference"). Test infrastructure used as the thing timed BESIDE the GPU.
# ---------------------------------------------------------------------------------------------------
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_pivots(budget_s=10.0):
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
    set_param(0, K) for K = 1 and K = 9 on the same LP; per-pivot time by differencing."""
    from oracle.checker import F64, Ref
    if not Ref.available():
        return None
    ref = Ref()
    ts = {}
    for K in (1, 9):                                     # eight pivots of signal (two gave a 2 x spread between runs: 14.7 ... 28.9 pivots/s)
        t0 = time.perf_counter()
        ref.two_stage(F64, leq, tgtf, K)
        ts[K] = time.perf_counter() - t0
    per = (ts[9] - ts[1]) / 8.0
    return dict(value=round(1.0 / per, 4) if per > 0 else None, unit="pivots/s", cores=1, kind="reference",
                sample="xcom::SIX<FloatMat,Float>::TwoStageMethod on the bench LP (m=4096,n=4095), "
                       "(t[K=9]-t[K=1])/8 = %.3f s/pivot; set-up %.1f s per call" % (per, ts[1] - per))


_POOL_STATE = {}


def _pool_worker(args):
    """One process per host core: solves its own slice of the problems, cyclically, for budget_s seconds."""
    kind, w, cores, budget_s = args
    from oracle.checker import RAT, Port
    port = Port()
    st = _POOL_STATE
    n, i = 0, w
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end:
        if kind == "lp":
            port.six_solve(0, True, st["tg"][i], st["vc"], None, st["leq"][i])
        elif kind == "fme":
            port.fme(st["leq"][i], st["tg"], 0, False)       # tg: the number of variables
        else:
            port.mip_solve(RAT, True, True, st["tg"][i], st["vc"], None, st["leq"][i])
        n += 1
        i += cores
        if i >= len(st["leq"]):
            i = w % len(st["leq"])
    return n


def cpu_per_core(kind, tg, vc, leq, budget_s, what):
    """The oracle on every host core at once: one PROCESS per core (forked before this process touches the
    GPU), each solving a disjoint slice for budget_s seconds. Returns items/s over all cores."""
    import multiprocessing as mp
    cores = host_cores()
    _POOL_STATE.update(tg=tg, vc=vc, leq=leq)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        counts = pool.map(_pool_worker, [(kind, w, cores, budget_s) for w in range(cores)], chunksize=1)
    dt = time.perf_counter() - t0
    n = sum(counts)
    return dict(value=round(n / budget_s, 2), cores=cores, kind="port",
                sample="%d %s through the oracle, one process per host core (%d), %.1f s each (%.1f s wall incl. fork)"
                       % (n, what, cores, budget_s, dt))


def cpu_baselines(legs, no_ref):
    """Every CPU figure of the run, taken BEFORE anything touches the GPU (the per-core legs fork)."""
    from oracle.checker import Port
    from tools import gen
    port = Port()
    cpu = {}
    if "pivots" in legs:
        cpu = cpu_baseline_pivots(8.0)
        if not no_ref:
            leq, tgtf = gen.hard_lp_f64(M, NVARS)
            cpu["reference"] = cpu_reference_pivots(leq, tgtf)
            del leq
    if "batched" in legs:
        leq_b, tg_b = gen.small_lp_batch_f64(2048, BATCH_M, BATCH_COLS, 1, seed=gen.XS_SEED + 1001)
        r = cpu_per_core("lp", tg_b, gen.vc_nonneg(BATCH_COLS - 1), leq_b, 6.0, "dep-test-like LPs (32x64, SIX::maxm)")
        r["unit"] = "LPs/s"
        cpu["batched"] = r
    if "rational" in legs:
        cpu["rational"] = cpu_rational(port, gen)
    if "mip" in legs:
        leq_m, tg_m = gen.knapsack_batch_rat(MIP_NB, MIP_NV)
        r = cpu_per_core("mip", tg_m, gen.to_rat(gen.vc_nonneg(MIP_NV, False)), leq_m, 4.0, "0-1 knapsack MIPs (%d vars)" % MIP_NV)
        r["unit"] = "MIPs/s"
        cpu["mip"] = r
    if "lineq" in legs:
        rng = np.random.default_rng(0)
        sysm = np.stack([gen.random_system(rng, 40, 12) for _ in range(512)])
        r = cpu_per_core("fme", 12, None, sysm, 3.0, "Lineq::fme eliminations (40x13 systems)")
        r["unit"] = "systems/s"
        cpu["lineq"] = r
    return cpu


# ---------------------------------------------------------------------------------------------------
# Self-checks: what the timed regions computed, compared AFTER the timing with fixtures generated from the
# real reference (tests/golden, tools/gen_golden*.py). A mismatch aborts the run: no JSON line, non-zero exit.
# ---------------------------------------------------------------------------------------------------
def checksum(a):
    import zlib
    a = np.ascontiguousarray(a)
    v = a.view(np.uint64).reshape(-1) if a.dtype.itemsize == 8 else a.view(np.uint32).reshape(-1).astype(np.uint64)
    return dict(crc32="%08x" % (zlib.crc32(a.tobytes()) & 0xFFFFFFFF), sum="%016x" % int(v.sum(dtype=np.uint64)),
                xor="%016x" % int(np.bitwise_xor.reduce(v)))


def golden(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name)))


def selfcheck_bench_lp(lp):
    """The device state the LAST timed step left (3840 pivots after xpg_lp_begin) against the real reference's
    TwoStageMethod(max_iter = 3840) on the same LP: whole tableau, objective row, basis."""
    rec = [r for r in golden("g11_bench_lp.json")["bench_lp"] if r["K"] == PIVOTS_PER_STEP][0]
    got = lp.read()
    bad = []
    if checksum(got["tab"]) != rec["tab"]:
        bad.append("tableau")
    if checksum(got["tgtf"]) != rec["tgtf"] or float(got["tgtf"][got["rhs"]]).hex() != rec["obj_const"]:
        bad.append("objective row")
    if checksum(got["eq2bv"].astype(np.int32)) != rec["eq2bv"] or checksum(got["bv2eq"].astype(np.int32)) != rec["bv2eq"]:
        bad.append("basis")
    if bad:
        sys.exit("bench.py self-check FAILED: after %d pivots the %s differ(s) from the reference fixture "
                 "tests/golden/g11_bench_lp.json" % (PIVOTS_PER_STEP, ", ".join(bad)))
    return dict(checked="tableau 4096x8192 (CRC-32 + sum + xor of all 33.5 M cells), objective row, eq2bv, bv2eq after the "
                        "last timed step's %d pivots" % PIVOTS_PER_STEP,
                against="tests/golden/g11_bench_lp.json: xcom::SIX<FloatMat,Float>::TwoStageMethod(max_iter=%d) of the real "
                        "reference on this LP (tools/gen_golden_bench.py)" % PIVOTS_PER_STEP,
                result="bit-identical")


def selfcheck_batched(fam, d_st, d_v, d_sol, gathered):
    """The first 256 LPs of rank 0's shard are the LPs of tests/golden/g8_large.json (g3_large: the real reference's
    SIX<FloatMat,Float>::maxm on each): status, objective bits and solution CRC of what the LAST timed pass left in
    the result arrays -- and, for N > 1, of the records the all_gather delivered."""
    import zlib
    rec = golden("g8_large.json")["g3_large"][fam]["records"]
    st = d_st[:256].cpu().numpy(); v = d_v[:256].cpu().numpy(); sol = d_sol[:256].cpu().numpy()
    for b, want in enumerate(rec):
        ok = st[b] == want["status"] and float(v[b]).hex() == want["v"]
        if ok and want["status"] == 0:
            ok = "%08x" % (zlib.crc32(np.ascontiguousarray(sol[b]).tobytes()) & 0xFFFFFFFF) == want["sol_crc32"]
        if ok and gathered is not None:
            g = gathered[b].cpu().numpy()
            ok = int(g[0]) == want["status"] and float(g[1]).hex() == want["v"]
        if not ok:
            sys.exit("bench.py self-check FAILED: batched LP %d of family %d differs from the reference fixture "
                     "tests/golden/g8_large.json (status %d v %s, want %s)" % (b, fam, st[b], float(v[b]).hex(), want))
    return "256 LPs of the timed batch (the g3_large fixture of tests/golden/g8_large.json, real reference): status, objective bits, solution CRC identical"


def spawn_ranks(a, argv):
    """--gpus N without a launcher: start N ranks of this script, before any GPU call, and exit with
    the launcher's code. A rank that cannot get its GPU fails loudly (XPG_ERR_NO_DEVICE) and the whole
    run exits non-zero -- never a silent 1-rank run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--legs", default=",".join(LEGS), help="comma-separated subset of " + ",".join(LEGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-baseline", action="store_true",
                    help="skip timing the real reference build (oracle/_ref, ~1 s on the GPU box's host)")
    ap.add_argument("--no-events", action="store_true", help="do not attach HIP events to sweep launches")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo + --stub-solver: the rank-spawn / shard / gather path without GPUs (tests)")
    ap.add_argument("--same-device", action="store_true",
                    help="test aid: every rank solves on cuda:0 and the collectives run over gloo on host tensors -- the real "
                         "solver through the whole N > 1 path (shards, records, gather) on a box with ONE GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 path -- rank spawn before any GPU call, init_process_group(nccl, device_id=...), device-resident "
                         "record tensors, all_gather_into_tensor -- also at --gpus 1: RCCL at world size 1 with the real solver")
    ap.add_argument("--stub-solver", action="store_true",
                    help="ranks fabricate their shard's records instead of solving (CPU test of the N > 1 path)")
    a = ap.parse_args()
    legs = set(x for x in a.legs.split(",") if x)
    assert legs <= set(LEGS), "unknown leg in --legs"

    if (a.gpus > 1 or a.force_dist) and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a, sys.argv[1:]))          # nothing has touched a GPU yet
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    stub = a.stub_solver
    cpu = None
    use_dist = world > 1 or a.force_dist
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not stub:
        cpu = cpu_baselines(legs, a.no_ref_baseline)     # forks: before torch / HIP are initialised in this process

    # the rank's host thread (and the pinned staging it allocates from here on) onto the NUMA node of its GPU -- from sysfs,
    # before anything in this process has touched a GPU (and after the CPU baseline has used every core)
    placement = None
    if not stub:
        from xpoly_amd.shard import pin_to_gpu_numa
        placement = pin_to_gpu_numa(local)
    import torch
    from xpoly_amd.shard import gather_records, pack_records, pack_records_i32, pack_records_rat, shard_range, unpack_records_rat
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.same_device:
            assert a.backend == "gloo", "--same-device needs --backend gloo (RCCL refuses two ranks on one GPU)"
            local = 0
        if stub or a.same_device:
            dist.init_process_group(backend=a.backend)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group(backend=a.backend, device_id=torch.device("cuda", local))
    dev = torch.device("cpu") if stub else torch.device("cuda", local if use_dist else 0)
    cdev = torch.device("cpu") if (stub or a.backend == "gloo") else dev     # where the collectives' tensors live
    if not stub:
        torch.cuda.set_device(dev)

    ctx = None
    if not stub:
        import xpoly_amd
        from tools import gen
        RUNNING = xpoly_amd.six.XPG_RUNNING
        ctx = xpoly_amd.Context(dev.index)              # XPG_ERR_NO_DEVICE -> exception -> non-zero exit

    def barrier():
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()
            ctx.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t)
        return float(t.item())

    out = {"metric": "simplex pivots/sec (float tableau 4kx8k)", "value": None, "unit": "pivots/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    if stub:
        out["stub_solver"] = True

    # ---- leg 1: pivots/s on the 4096 x 8192 tableau (one replica per rank) -------------------------
    leq = tgtf = None
    if "pivots" in legs and not stub:
        leq, tgtf = gen.hard_lp_f64(M, NVARS)          # the same LP on every rank (identical replicas)
        lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tgtf)

        def run_step():
            lp.begin()                                  # slack form rebuilt on the device (one kernel)
            st = lp.iterate(PIVOTS_PER_STEP)
            assert st == RUNNING, "LP ended early (status %d)" % st

        # set-up, not measured: one pass through every code path of the timed region (launch throttling,
        # event pairs) so lazy runtime initialisation does not land inside it. The ROCm runtime was measured
        # to stall the stream once for 60-80 ms somewhere in the first ~100 ms of queued-loop activity of a
        # process (tools/lab/probe_stall.py), so the set-up pass runs for at least PREWARM_SECONDS of wall time.
        ctx.profile_begin(8, 32)
        t_pre = time.perf_counter()
        run_step()
        ctx.sync()
        while time.perf_counter() - t_pre < PREWARM_SECONDS:
            run_step()
            ctx.sync()
        ctx.profile_end()
        for _ in range(a.warmup):
            run_step()
        barrier()
        batches = a.steps * (PIVOTS_PER_STEP // BLOCK)
        stride = max(1, batches // 320)                 # ~320 sampled sweep launches spread over the region
        ctx.profile_begin(0 if a.no_events else batches // stride + 1, stride)
        sweeps_full = sweeps_part = 0
        pivots = 0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run_step()
            if not a.no_events:                         # device counters of this step (the iterate call has
                f, p = lp.counters()                    # synchronised already: no extra round trip inside)
                sweeps_full += f
                sweeps_part += p
            pivots += PIVOTS_PER_STEP
        ctx.sync()
        barrier()
        dt = time.perf_counter() - t0
        launches, sweep_ms = ctx.profile_end()
        assert lp.pivots_done() % PIVOTS_PER_STEP == 0
        rows, W, rhs = lp.shape()
        assert (rows, W) == (M, TAB_W)
        dt = max_over_ranks(dt)
        value = world * pivots / dt
        out["value"] = round(value, 2)
        out["ms_per_step"] = round(dt / a.steps * 1e3, 4)
        out["config"] = {
            "workload": "dense LP m=4096 n=4095 (gen.hard_lp_f64: A~U(0.1,1), b=A x*, c=A^T y*), slack tableau "
                        "4096x8192 fp64 resident in HBM; one step = xpg_lp_begin (slack form rebuilt on the device) "
                        "+ %d pivots of the device-resident SIX::solveSlackForm loop = %d full batches of %d"
                        % (PIVOTS_PER_STEP, PIVOTS_PER_STEP // BLOCK, BLOCK),
            "tableau": [M, TAB_W], "pivots_per_step": PIVOTS_PER_STEP, "timed_region_ms": round(dt * 1e3, 2),
            "parallelism": "replicas only (1 tableau per GPU)"}
        if launches:
            sweep_avg_s = sweep_ms / 1e3 / launches
            achieved = ALG_BYTES_PER_LAUNCH / sweep_avg_s / 1e9
            n_sweeps = sweeps_full + sweeps_part
            ppl = pivots / n_sweeps if n_sweeps else float(BLOCK)
            traffic, traffic_src = None, None
            if os.path.exists(PMC_SUMMARY):             # measured with rocprofv3 --pmc (separate passes), not live
                pm = json.load(open(PMC_SUMMARY))
                traffic = round(pm["traffic_bytes_per_launch"])
                traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes of this "
                               "loop; not collected in this run)" % os.path.basename(PMC_SUMMARY))
            per_pivot_s = dt / pivots
            out["roofline"] = dict(
                bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src,
                kernel="k_blk_sweep_full<%s,%d>: one launch applies the %d staged pivots of a batch to every cell (XCD-static "
                       "tile map, alternate passes in opposite directions)" % ("16,2" if BLOCK in (24, 32) else "16,4", BLOCK, BLOCK),
                algorithmic_bytes_per_launch=ALG_BYTES_PER_LAUNCH, launches_sampled=launches,
                avg_launch_us=round(sweep_avg_s * 1e6, 2),
                sweeps_in_region=dict(full=sweeps_full, partial=sweeps_part),
                pivots_per_launch=round(ppl, 3),
                loop_effective=dict(
                    note="whole loop: bytes the sweeps move per pivot / wall time per pivot; the pick and prep "
                         "launches between sweeps move next to nothing, so this is what the loop as a whole "
                         "draws from HBM",
                    bytes_per_pivot=round(ALG_BYTES_PER_LAUNCH / ppl), us_per_pivot=round(per_pivot_s * 1e6, 3),
                    achieved=round(ALG_BYTES_PER_LAUNCH / ppl / per_pivot_s / 1e9, 1),
                    frac=round(ALG_BYTES_PER_LAUNCH / ppl / per_pivot_s / 1e9 / HBM_PEAK_GBS, 4)),
                caveat="fraction of the 8 TB/s HBM peak for ONE pass that applies %d staged pivots to every cell (per pivot the "
                       "pass moves 1/%d of these bytes). The 268 MB tableau is exactly the size of the 256 MiB Infinity Cache "
                       "(MALL) and FETCH_SIZE counts its hits, so part of the stream is served by the MALL. The same kernel on a "
                       "tableau 1.57 x the MALL is the `cfg2b.roofline` object of this line; a plain in-place copy with the same "
                       "tiling runs at 0.86 / 0.79 of the peak at the two sizes (profiles/round3_sweep_lab.txt). At 32 stages the pass is "
                       "bound by fp64 issue next to memory: 2 x 33.5 M x 32 non-fused operations in 93.5 us are 0.62 of the 37.2 T "
                       "lane-operations/s the device issues of v_mul_f64 / v_add_f64 alone (tools/lab/valu_f64_lab.hip) and 0.74 of the 31 T/s "
                       "at the 1.9-1.97 GHz it holds UNDER this pass (s_memtime inside it, tools/lab/sweep_lab3.hip). With XPG_BLOCK=24 "
                       "the pass is at 0.81 of the HBM peak (81 us) and with 16 at the copy ceiling (78 us, 0.86), but those bytes "
                       "are paid per 24 / 16 pivots: 128.7 k / 107.2 k pivots/s against 134.4 k with 32 "
                       "(profiles/round5_block_length_ab.txt, DESIGN section 4.2c)" % (BLOCK, BLOCK),
                chain=dict(note="the time-dominant kernel of the loop is not the pass but k_blk_chain, the persistent launch that "
                                "stages the batch's pivots: latency-bound (two L2 hand-offs and two gathers per stage), no byte or "
                                "flop roofline applies; us per stage from the whole-loop time",
                           us_per_stage=round((per_pivot_s * 1e6 * ppl - sweep_avg_s * 1e6) / ppl, 3) if ppl else None))
        if rank == 0:
            out["self_check"] = {"pivots": selfcheck_bench_lp(lp)}      # outside the timed region
        lp.close()

    # ---- leg 2: batched 32 x 64 LPs, sharded across ranks, one all_gather at the end ----------------
    b_leq = b_tg = None
    checks = {}
    if "batched" in legs:
        total = BATCH_PER_GPU * world
        lo, hi = shard_range(total, rank, world)
        nloc = hi - lo
        fams = {}
        h2d_ms = {}
        for fam, name in ((1, "dep_test_like"), (0, "dense_positive")):
            if stub:
                idx = torch.arange(lo, hi, dtype=torch.float64)
                d_st = (idx % 5).to(torch.int32)
                d_v = idx * 0.5 + fam
                d_sol = idx[:, None] + torch.arange(BATCH_COLS, dtype=torch.float64)[None, :] / 100.0
                d_piv = torch.ones(nloc, dtype=torch.int32)
            else:
                b_leq, b_tg = gen.small_lp_batch_f64(nloc, BATCH_M, BATCH_COLS, fam,
                                                     seed=gen.XS_SEED + 1000 * (rank + 1) + fam)
                if rank == 0 and nloc >= 256:           # the 256 LPs of the reference fixture lead rank 0's shard: checked after the timing
                    g_rec = golden("g8_large.json")["g3_large"][fam]
                    g_leq, g_tg = gen.small_lp_batch_f64(256, BATCH_M, BATCH_COLS, fam, seed=gen.XS_SEED + g_rec["seed_offset"])
                    b_leq[:256] = g_leq; b_tg[:256] = g_tg
                torch.cuda.synchronize()
                t_up = time.perf_counter()
                d_leq = torch.from_numpy(b_leq).to(dev)
                d_tg = torch.from_numpy(b_tg).to(dev)
                torch.cuda.synchronize()
                h2d_ms[name] = round((time.perf_counter() - t_up) * 1e3, 3)     # the shard's upload, once, outside the timed passes
                d_st = torch.empty(nloc, dtype=torch.int32, device=dev)
                d_v = torch.empty(nloc, dtype=torch.float64, device=dev)
                d_sol = torch.zeros(nloc, BATCH_COLS, dtype=torch.float64, device=dev)
                d_piv = torch.empty(nloc, dtype=torch.int32, device=dev)

            tsplit = [0.0, 0.0]                            # seconds in the solve / in the gather, this rank

            def one_pass():
                ta = time.perf_counter()
                if not stub:
                    ctx.six_batch_dev(xpoly_amd.F64, True, nloc, d_tg.data_ptr(), d_leq.data_ptr(),
                                      BATCH_M, BATCH_COLS, d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(),
                                      d_piv.data_ptr())
                    ctx.sync()
                tb = time.perf_counter()
                tsplit[0] += tb - ta
                if dist is None:
                    return None
                # the only collective of the path: fixed-size (status, v, sol) records
                full = gather_records(pack_records(d_st, d_v, d_sol).to(cdev), total, rank, world, dist)
                if not stub:
                    torch.cuda.synchronize()
                tsplit[1] += time.perf_counter() - tb
                return full

            full = one_pass()
            barrier()
            reps = 3 if fam == 1 else 5
            tsplit[0] = tsplit[1] = 0.0
            t0 = time.perf_counter()
            for _ in range(reps):
                full = one_pass()
            barrier()
            bdt = max_over_ranks(time.perf_counter() - t0)
            solve_ms = max_over_ranks(tsplit[0]) / reps * 1e3
            gather_ms = max_over_ranks(tsplit[1]) / reps * 1e3
            if full is not None:
                assert full.shape[0] == total
                if stub:                                # the gathered order is the global LP order
                    gi = torch.arange(total, dtype=torch.float64)
                    assert torch.equal(full[:, 0], gi % 5) and torch.equal(full[:, 1], gi * 0.5 + fam)
                    assert torch.equal(full[:, 2], gi)
            piv = sum_over_ranks(float(d_piv.sum().item()))
            hist = torch.bincount(d_st.clamp(min=0), minlength=5).tolist()
            # ---- what a HOST caller sees: the shard's arrays start in (pageable) host memory. Once as upload-then-solve, once
            # with the upload in two chunks (a quarter, then the rest) on a side stream, the second under the first chunk's solve
            e2e = None
            if not stub:
                import threading

                def upload(lo, hi, side):
                    with torch.cuda.stream(side):
                        dl = torch.from_numpy(b_leq[lo:hi]).to(dev)
                        dtg = torch.from_numpy(b_tg[lo:hi]).to(dev)
                    side.synchronize()
                    return dl, dtg

                def e2e_pass(chunks):
                    side = torch.cuda.Stream()
                    torch.cuda.synchronize(); ctx.sync()
                    t0 = time.perf_counter()
                    cur = upload(chunks[0][0], chunks[0][1], side)
                    for k, (lo, hi) in enumerate(chunks):
                        box, th = [], None
                        if k + 1 < len(chunks):
                            th = threading.Thread(target=lambda c=chunks[k + 1]: box.append(upload(c[0], c[1], side)))
                            th.start()
                        ctx.six_batch_dev(xpoly_amd.F64, True, hi - lo, cur[1].data_ptr(), cur[0].data_ptr(), BATCH_M, BATCH_COLS,
                                          d_st[lo:hi].data_ptr(), d_v[lo:hi].data_ptr(), d_sol[lo:hi].data_ptr(), d_piv[lo:hi].data_ptr())
                        ctx.sync()
                        if th is not None:
                            th.join()
                            cur = box[0]
                    return time.perf_counter() - t0

                want_st, want_v = d_st.clone(), d_v.clone()
                q = max(1, nloc // 4)
                plans = {"upload_then_solve": [(0, nloc)], "two_chunks_overlapped": [(0, q), (q, nloc)] if nloc > q else [(0, nloc)]}
                e2e = {}
                for pname, chunks in plans.items():
                    best = min(e2e_pass(chunks) for _ in range(2))
                    assert torch.equal(d_st, want_st) and torch.equal(d_v.view(torch.int64), want_v.view(torch.int64)), "chunked solve differs"
                    e2e[pname + "_lps_per_s"] = round(total / max_over_ranks(best), 1)
                # ... and the host-array entry point itself (xpg_six_batch_f64: host arrays in AND out, one upload, one launch)
                best = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    hst, hv, hsol = ctx.six_batch(xpoly_amd.F64, True, b_tg, b_leq)
                    dt_h = time.perf_counter() - t0
                    best = dt_h if best is None else min(best, dt_h)
                assert np.array_equal(hst, want_st.cpu().numpy()) and np.array_equal(hv.view(np.int64), want_v.cpu().numpy().view(np.int64)), "host-array call differs"
                e2e["host_array_call_lps_per_s"] = round(total / max_over_ranks(best), 1)
                e2e["note"] = ("host arrays in (pageable): upload_then_solve / two_chunks_overlapped leave the results in HBM (torch uploads + the "
                               "device-array entry point; two LAUNCHES are slower than one: each has a tail of its own); host_array_call is "
                               "xpg_six_batch_f64 -- results back on the host too. An upload UNDER one launch (chunks on a second stream, "
                               "workgroups waiting on a gate word) was built in round 5 and deadlocks on a full chip: the runtime's copies "
                               "of pageable memory are shader copies that find no registers while every seat spins (DESIGN section 5); "
                               "best of the passes; the device-resident lps_per_s beside them excludes the upload")
            if rank == 0 and not stub and nloc >= 256:
                checks.setdefault("batched", {})[name] = selfcheck_batched(fam, d_st, d_v, d_sol, full)
            fams[name] = dict(lps_per_s=round(total * reps / bdt, 1), pivots_per_s=round(piv * reps / bdt, 1),
                              status_hist_rank0=hist, ms_per_pass=round(bdt / reps * 1e3, 3),
                              solve_ms=round(solve_ms, 3), gather_ms=round(gather_ms, 3),
                              shard_h2d_ms_untimed=h2d_ms.get(name), end_to_end=e2e)
        batched = dict(metric="batched LPs/sec", value=fams["dep_test_like"]["lps_per_s"], unit="LPs/s",
                       headline_family="dep_test_like (entries in {-3..3} at density 0.25: the shape "
                                       "DepPoly::is_empty produces; the workload the kernel exists for)",
                       n_gpus=world, ranks=world, total_lps=total, lps_per_rank=[shard_range(total, r, world)[1] -
                                                                                 shard_range(total, r, world)[0]
                                                                                 for r in range(world)],
                       shape="leq 32x64 (63 vars + rhs), SIX::maxm, x>=0, whole solve per LP in LDS; LPs that have run their time slice (512 iterations) hand their LDS seat back and continuation workgroups inside the launch take queued LPs in turn (results do not depend on it)",
                       scaling="weak (8192 LPs per GPU: 65 536 at N = 8)",
                       collective=("one all_gather_into_tensor of (status,v,sol) records over %s, world size %d"
                                   % ("RCCL" if a.backend == "nccl" else a.backend, world)) if dist else "none (1 GPU)",
                       families=fams)
        if os.path.exists(PMC_BATCH):                   # measured with rocprofv3 --pmc, not in this run
            pb = json.load(open(PMC_BATCH))
            batched["issue_rate"] = dict(
                bound="latency of the two-barrier pivot at 5 LPs per CU (LDS: 30 KB per LP; the pivot loop is a function of its own on 88 registers, the kernel is held to 96 to seat the fifth LP); with the time slices every seat stays busy to the end of a launch: VALU 48-52 % and SALU 23-25 % busy over the whole launch (HBM sees 16 KiB in / 0.5 KiB out per LP)",
                dep_test_like=dict(valu_busy_percent=pb.get("dep_test_like_busy_percent", {}).get("VALUBusy"),
                                   salu_busy_percent=pb.get("dep_test_like_busy_percent", {}).get("SALUBusy"),
                                   wave_instructions_per_pivot=pb.get("dep_test_like_per_pivot")),
                dense_positive=dict(valu_busy_percent=pb.get("dense_busy_percent", {}).get("VALUBusy"),
                                    salu_busy_percent=pb.get("dense_busy_percent", {}).get("SALUBusy"),
                                    wave_instructions_per_pivot=pb.get("dense_per_pivot")),
                source="profiles/round5_pmc_batch_issue.json (rocprofv3 --pmc VALUBusy / SALUBusy / SQ_INSTS_* on "
                       "tools/lab/probe_batch.py; not collected in this run)")
        if not stub:
            # STRONG scaling beside the weak figures above: the 65 536 LPs of BASELINE configs[2] in all, split over the
            # ranks (65 536 / N each), same solve + one all_gather; at N = 1 this is the whole batch on one GPU
            full_n = BATCH_PER_GPU * 8
            slo, shi = shard_range(full_n, rank, world)
            sn = shi - slo
            strong = {}
            for fam, name in ((1, "dep_test_like"), (0, "dense_positive")):
                f_leq, f_tg = gen.small_lp_batch_f64(sn, BATCH_M, BATCH_COLS, fam, seed=gen.XS_SEED + 77 + fam + 131 * rank)
                d_leq = torch.from_numpy(f_leq).to(dev); d_tg = torch.from_numpy(f_tg).to(dev)
                d_st = torch.empty(sn, dtype=torch.int32, device=dev)
                d_v = torch.empty(sn, dtype=torch.float64, device=dev)
                d_sol = torch.zeros(sn, BATCH_COLS, dtype=torch.float64, device=dev)
                best = None
                for rep in range(2):
                    barrier()
                    t0 = time.perf_counter()
                    ctx.six_batch_dev(xpoly_amd.F64, True, sn, d_tg.data_ptr(), d_leq.data_ptr(), BATCH_M, BATCH_COLS,
                                      d_st.data_ptr(), d_v.data_ptr(), d_sol.data_ptr(), None)
                    ctx.sync()
                    if dist is not None:
                        g = gather_records(pack_records(d_st, d_v, d_sol).to(cdev), full_n, rank, world, dist)
                        torch.cuda.synchronize()
                        assert g.shape[0] == full_n
                    barrier()
                    dt_s = max_over_ranks(time.perf_counter() - t0)
                    best = dt_s if best is None else min(best, dt_s)
                strong[name] = round(full_n / best, 1)
                del d_leq, d_tg, d_st, d_v, d_sol, f_leq, f_tg
            batched["strong_scaling"] = dict(total_lps=full_n, lps_per_rank=sn, lps_per_s=strong,
                                             note="fixed total of 65 536 LPs split over the ranks, solve + all_gather, better of two passes")
        out["batched"] = batched

    # ---- leg 2b: the exact (rational) batches of BASELINE configs[4], sharded like leg 2 --------------------------
    if "sharded" in legs:
        out["sharded"] = leg_sharded(ctx, rank, world, cdev, dist, stub, a.backend, barrier, max_over_ranks, sum_over_ranks)

    # ---- legs 3-5: the remaining BASELINE configs, rank 0 of an N = 1 run ---------------------------
    if world == 1 and not stub:
        if "cfg2b" in legs:
            out["cfg2b"] = leg_cfg2b(ctx, xpoly_amd, gen)
        if "shapes" in legs:
            out["shapes"] = leg_shapes(ctx, xpoly_amd, gen)
        if "six_e2e" in legs:
            out["six_e2e"] = leg_six_e2e(ctx, xpoly_amd, gen, with_reference=not a.no_cpu_baseline and not a.no_ref_baseline)
        if "one_call" in legs:
            out["one_call"] = leg_one_call(ctx, xpoly_amd, gen, with_reference=not a.no_cpu_baseline and not a.no_ref_baseline)
        if "rational" in legs:
            out["rational"] = leg_rational(ctx, xpoly_amd, gen)
        if "mip" in legs:
            out["mip"] = leg_mip(ctx, xpoly_amd, gen)
        if "lineq" in legs:
            out["lineq"] = leg_lineq(ctx, xpoly_amd, gen)

    if world == 1 and dist is None and not stub and "batched" in legs and "batched" in out:
        out["batched"]["rccl_world1_fixed_cost"] = rccl_world1_fixed_cost(torch, dev)
    if rank == 0 and placement is not None:
        out["host_placement"] = dict(placement, note="rank 0: NUMA node of its GPU from sysfs, affinity set before the first GPU call (xpoly_amd/shard.py pin_to_gpu_numa)")
    if rank == 0:
        if checks or "self_check" in out or any("self_check" in out.get(leg, {}) for leg in ("mip", "cfg2b", "rational")):
            out.setdefault("self_check", {}).update(checks)
            if "mip" in out and "self_check" in out["mip"]:
                out["self_check"]["mip"] = out["mip"].pop("self_check")
            for leg in ("cfg2b", "rational"):
                if leg in out and "self_check" in out[leg]:
                    out["self_check"][leg] = out[leg].pop("self_check")
        out["cpu_baseline"] = cpu
        if cpu is None and world > 1:
            out["cpu_baseline_note"] = "timed on rank 0 of the N = 1 run only (the contract's rule); not repeated at N > 1"
        print(json.dumps(out))
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def rccl_world1_fixed_cost(torch, dev):
    """The collective's fixed cost on this box, on file in every N = 1 line: RCCL at world size 1 (no launcher: a TCP store on
    127.0.0.1), one all_gather_into_tensor of each record shape the sharded legs gather -- 8192 LP records of 528 B, 1024 exact
    MIP records, 4096 dependence verdicts -- mean of 10 after 3 warm-ups. An 8-GPU run adds the ring over xGMI to this; nothing
    here claims that figure. Never fatal: an RCCL that cannot initialise is reported as such."""
    try:
        import torch.distributed as dist
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        t_init = time.perf_counter()
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        res = {}
        shapes = (("lp_records_8192x66_f64", (8192, 66), torch.float64), ("mip_records_1024x53_i32", (1024, 53), torch.int32),
                  ("dep_verdicts_4096x2_i32", (4096, 2), torch.int32))
        for name, shape, dtype in shapes:
            src = torch.zeros(shape, dtype=dtype, device=dev)
            dst = torch.empty_like(src)
            for _ in range(3):
                dist.all_gather_into_tensor(dst, src)
            torch.cuda.synchronize()
            if "init_ms" not in res:
                res["init_ms"] = round((time.perf_counter() - t_init) * 1e3, 1)      # communicator set-up incl. the first collective
            t0 = time.perf_counter()
            for _ in range(10):
                dist.all_gather_into_tensor(dst, src)
            torch.cuda.synchronize()
            res[name + "_gather_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
        dist.destroy_process_group()
        res["note"] = "RCCL all_gather_into_tensor at world size 1 on this GPU; the sharded legs' one collective, fixed cost only"
        return res
    except Exception as e:                                  # noqa: BLE001 -- evidence only, never the reason a bench line is lost
        return {"error": "%s: %s" % (type(e).__name__, e)}


# ---------------------------------------------------------------------------------------------------
SH_MIP_PER_GPU, SH_DEP_PER_GPU = 1024, 4096


def leg_sharded(ctx, rank, world, dev, dist, stub, backend, barrier, max_over_ranks, sum_over_ranks):
    """BASELINE configs[4] as the north star states it -- independent exact problems sharded over the GPUs of the node,
    no data-path collective, ONE gather of fixed-size records at the end -- for the two rational batch paths:
      mip           0-1 knapsack MIPs (24 variables), xpg_mip_batch_rat32: each rank walks its own trees on its GPU;
                    records (status, v num/den, sol num/den) gathered as int32 (exact: nothing goes through floats)
      dep_is_empty  dependence polyhedra 12 x 5, xpg_dep_is_empty_batch_rat32: verdicts gathered as int32
    Weak scaling: SH_MIP_PER_GPU / SH_DEP_PER_GPU problems per rank. Host arrays in and out on every rank (PCIe
    included), as an xpoly caller at linsys.cpp:860-904 would use it. Solve and gather are timed separately."""
    import torch
    from xpoly_amd.shard import gather_records, pack_records_i32, pack_records_rat, shard_range, unpack_records_rat
    res = {}
    for what, per_gpu in (("mip", SH_MIP_PER_GPU), ("dep_is_empty", SH_DEP_PER_GPU)):
        total = per_gpu * world
        lo, hi = shard_range(total, rank, world)
        nloc = hi - lo
        if stub:
            idx = np.arange(lo, hi, dtype=np.int64)
        elif what == "mip":
            from tools import gen
            from xpoly_amd.six import mip_batch
            leq, tgtf = gen.knapsack_batch_rat(nloc, MIP_NV, seed=gen.XS_SEED + 7919 * (rank + 1))
        else:
            from tools import gen
            from xpoly_amd.six import dep_is_empty_batch
            rng = np.random.default_rng(1000 + rank)
            dm = np.stack([gen.random_system(rng, 12, 4) for _ in range(256)])
            dm[..., 1] = 1
            dm = np.ascontiguousarray(np.tile(dm, (nloc // 256 + 1, 1, 1, 1))[:nloc])

        def solve():
            if stub:
                if what == "mip":
                    st = (idx % 4).astype(np.int32)
                    v = np.stack([idx * 3 + 1, np.ones_like(idx)], axis=1).astype(np.int32)
                    sol = (idx[:, None, None] + np.arange(MIP_NV + 1)[None, :, None] * 2 + np.arange(2)[None, None, :]).astype(np.int32)
                    return pack_records_rat(st, v, sol), int(nloc)
                return pack_records_i32((idx % 3).astype(np.int32), (idx * 7 % 1000).astype(np.int32)), int(nloc)
            if what == "mip":
                st, v, sol, nodes = mip_batch(ctx, True, True, tgtf, leq)
                return pack_records_rat(st, v, sol), int(nodes)
            empty, nodes = dep_is_empty_batch(ctx, dm)
            return pack_records_i32(empty), int(nodes)

        def one_pass():
            ta = time.perf_counter()
            rec, nodes = solve()
            tb = time.perf_counter()
            full = None
            if dist is not None:
                rec = rec.to(dev)                           # the records go up (a few hundred KB), then ONE all_gather
                full = gather_records(rec, total, rank, world, dist)
                if not stub:
                    torch.cuda.synchronize()
            return rec, full, nodes, tb - ta, time.perf_counter() - tb
        one_pass()
        barrier()
        t0 = time.perf_counter()
        rec, full, nodes, ts, tg = one_pass()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        if full is not None:
            assert full.shape[0] == total and full.dtype == torch.int32
            if stub:                                        # the gathered records are the global problem order, exactly
                gi = torch.arange(total, dtype=torch.int64)
                if what == "mip":
                    st, v, sol = unpack_records_rat(full.cpu())
                    assert torch.equal(st.to(torch.int64), gi % 4) and torch.equal(v[:, 0].to(torch.int64), gi * 3 + 1)
                    assert torch.equal(sol[:, 5, 1].to(torch.int64), gi + 11)
                else:
                    assert torch.equal(full.cpu()[:, 0].to(torch.int64), gi % 3)
        res[what] = dict(problems_total=total, problems_per_rank=nloc, problems_per_s=round(total / dt, 1),
                         nodes_per_s=round(sum_over_ranks(float(nodes)) / dt, 1), wall_ms=round(dt * 1e3, 3),
                         solve_ms=round(max_over_ranks(ts) * 1e3, 3), gather_ms=round(max_over_ranks(tg) * 1e3, 3),
                         record_int32s=int(rec.shape[1]))
    res["scaling"] = "weak (%d MIPs and %d polyhedra per GPU)" % (SH_MIP_PER_GPU, SH_DEP_PER_GPU)
    res["collective"] = ("one all_gather_into_tensor of int32 records over %s, world size %d" % ("RCCL" if backend == "nccl" else backend, world)) if dist else "none (1 GPU)"
    res["ranks"] = world
    return res


def leg_cfg2b(ctx, xpoly_amd, gen, m=4096, n=8192):
    """BASELINE configs[1] end to end: LP m=4096, n=8192 -> slack tableau 4096 x 12289 (lpsol.h:1406-1433).
    The SURVEY 8d recipe (b = n U(0.5,1)) reaches its optimum in a few dozen pivots, too few to time, so the
    LP is gen.hard_lp_f64 at this size (same A, optimum with every structural variable basic); rate =
    (K2 - K1) / (t2 - t1) with K1 = 256, K2 = 1280."""
    RUNNING = xpoly_amd.six.XPG_RUNNING
    leq, tgtf = gen.hard_lp_f64(m, n)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tgtf)
    del leq
    W = n + m + 1
    launches = sweep_ms = 0
    pers = []
    for rep in range(3):                                # the first pass warms every launch path; the rate is the better of the
        lp.begin()                                      # other two (the runtime was seen to stall a stream once for tens of ms)
        if rep == 2:
            ctx.profile_begin(40, 2)                    # HIP events on every other full-batch sweep launch of the last pass
        t0 = time.perf_counter()
        assert lp.iterate(256) == RUNNING
        t1 = time.perf_counter()
        assert lp.iterate(1024) == RUNNING
        t2 = time.perf_counter()
        if rep == 2:
            launches, sweep_ms = ctx.profile_end()
        if rep:
            pers.append((t2 - t1) / 1024.0)
    per = min(pers)
    f, p = lp.counters()
    # self-check, outside the timing: the state the last pass left (256 + 1024 pivots) against the real reference
    check = None
    recs = [r for r in golden("g11_bench_lp.json").get("cfg2b_lp", []) if r["K"] == 1280]
    if recs and (m, n) == (4096, 8192):
        got = lp.read()
        bad = [k for k, ok in (("tableau", checksum(got["tab"]) == recs[0]["tab"]), ("objective row", checksum(got["tgtf"]) == recs[0]["tgtf"]),
                               ("basis", checksum(got["eq2bv"].astype(np.int32)) == recs[0]["eq2bv"])) if not ok]
        if bad:
            sys.exit("bench.py self-check FAILED: cfg2b after 1280 pivots, %s differ(s) from tests/golden/g11_bench_lp.json" % ", ".join(bad))
        check = ("tableau 4096x12289 (CRC-32 + sum + xor), objective row, basis after the last pass's 1280 pivots = the real reference's "
                 "TwoStageMethod(max_iter=1280) on this LP (tests/golden/g11_bench_lp.json cfg2b_lp): bit-identical")
        del got
    lp.close()
    bytes_per_launch = 2 * m * W * 8
    out = dict(metric="simplex pivots/sec, LP m=4096 n=8192 (tableau 4096x12289 fp64)", value=round(1.0 / per, 1),
               unit="pivots/s", us_per_pivot=round(per * 1e6, 3), tableau=[m, W],
               algorithmic_bytes_per_launch=bytes_per_launch, pivots_per_launch=BLOCK,
               sweeps=dict(full=f, partial=p),
               loop_effective_gbs=round(bytes_per_launch / BLOCK / per / 1e9, 1),
               loop_effective_frac=round(bytes_per_launch / BLOCK / per / 1e9 / HBM_PEAK_GBS, 4),
               sample="(t[K=1280] - t[K=256]) / 1024 on one LP, device-resident blocked loop, better of two passes")
    if check:
        out["self_check"] = check
    if launches:
        avg = sweep_ms / 1e3 / launches
        out["roofline"] = dict(
            bound="hbm", kernel="k_blk_sweep_full<%s,%d> on the 403 MB tableau (1.57 x the 256 MiB Infinity Cache)" % ("32,2" if BLOCK == 32 else "16,2" if BLOCK == 24 else "16,4", BLOCK),
            achieved=round(bytes_per_launch / avg / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(bytes_per_launch / avg / 1e9 / HBM_PEAK_GBS, 4), avg_launch_us=round(avg * 1e6, 2),
            launches_sampled=launches,
            traffic=(round(json.load(open(PMC_SUMMARY_2B))["traffic_bytes_per_launch"]) if os.path.exists(PMC_SUMMARY_2B) else None),
            traffic_source="profiles/round6_pmc_hbm_traffic_4096x12289.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this leg; not collected in this run)",
            note="HIP events on the sweep launches of the timed pass; a plain in-place copy of this tableau with the same "
                 "tiling runs at 0.79 of the peak (tools/lab/sweep_lab2.hip, profiles/round3_sweep_lab.txt)")
    return out


SHAPES = (("tall", 16384, 2048), ("wide", 1024, 20480), ("square", 8192, 8192), ("odd_width", 4096, 4094), ("small", 2048, 2047))
SHAPE_K_CHECK, SHAPE_K_TIMED = 256, 512


def leg_shapes(ctx, xpoly_amd, gen):
    """Is the headline a tuning to the 4096 x 8192 stride? gen.hard_lp_f64(m, n) at other LP shapes -- tall (16384 x 2048:
    tableau 16384 x 18433, 2.4 GB), wide (1024 x 20480: tableau 1024 x 21505 -- the reference's own range ends where (n + m)^2 x 8 B wraps 2^32: its vc matrix; n + m >= 23 171 corrupts its heap), square (8192 x 8192: tableau 8192 x 16385), a
    tableau of odd width (4096 x 8191) and a small one (2048 x 4096) -- through the same automatic loop choice. Per shape:
    the state after 256 pivots is checked against the REAL reference (tests/golden/g13_shapes.json) before anything is
    timed; then (t[K = 768] - t[K = 256]) / 512 pivots, HIP events on the full-batch passes, the chain's share from the
    difference, and which loop / chain / pass instance ran (xpg_lp_loop_info)."""
    RUNNING = xpoly_amd.six.XPG_RUNNING
    gold = golden("g13_shapes.json")
    rows = []
    for name, m, n in SHAPES:
        leq, tgtf = gen.hard_lp_f64(m, n)
        lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.F64, leq, tgtf)
        del leq
        W = n + m + 1
        lp.begin()
        st = lp.iterate(SHAPE_K_CHECK)
        rec = dict(shape=name, lp=[m, n], tableau=[m, W], tableau_mb=round(m * W * 8 / 1e6, 1))
        if st != RUNNING:
            rec["ended_early_status"] = int(st)
            rows.append(rec); lp.close(); continue
        g = gold.get(name)
        if g:                                           # the verified state: whole tableau, objective row, basis at K = 256
            got = lp.read()
            bad = [k for k, ok in (("tableau", checksum(got["tab"]) == g["tab"]), ("objective row", checksum(got["tgtf"]) == g["tgtf"]),
                                   ("basis", checksum(got["eq2bv"].astype(np.int32)) == g["eq2bv"])) if not ok]
            del got
            if bad:
                sys.exit("bench.py self-check FAILED: shape %s after 256 pivots, %s differ(s) from tests/golden/g13_shapes.json" % (name, ", ".join(bad)))
            rec["self_check"] = "state after 256 pivots = the real reference's (tests/golden/g13_shapes.json): bit-identical"
        else:
            rec["self_check"] = None
        info = lp.loop_info()
        pers, launches, sweep_ms = [], 0, 0.0
        for rep in range(3):
            lp.begin()
            assert lp.iterate(SHAPE_K_CHECK) == RUNNING
            if rep == 2:
                ctx.profile_begin(32, 1)
            t1 = time.perf_counter()
            st = lp.iterate(SHAPE_K_TIMED)
            t2 = time.perf_counter()
            if rep == 2:
                launches, sweep_ms = ctx.profile_end()
            if rep and st == RUNNING:
                pers.append((t2 - t1) / SHAPE_K_TIMED)
        lp.close()
        if not pers:
            rec["ended_early_status"] = int(st)
            rows.append(rec); continue
        per = min(pers)
        rec.update(pivots_per_s=round(1.0 / per, 1), us_per_pivot=round(per * 1e6, 3), loop=info)
        alg = 2 * m * W * 8
        if launches and info["loop"] == "blocked":
            avg = sweep_ms / 1e3 / launches
            ppl = info["pivots_per_pass"]
            rec["pass"] = dict(kernel="k_blk_sweep_full<%d,2,%d>" % (info["sweep_rows"], ppl), avg_launch_us=round(avg * 1e6, 2), launches_sampled=launches,
                               algorithmic_bytes_per_launch=alg, achieved_gbs=round(alg / avg / 1e9, 1), frac_of_hbm_peak=round(alg / avg / 1e9 / HBM_PEAK_GBS, 4))
            rec["chain_us_per_stage"] = round((per * 1e6 * ppl - avg * 1e6) / ppl, 3)
            rec["loop_effective_frac"] = round(alg / ppl / per / 1e9 / HBM_PEAK_GBS, 4)
        rows.append(rec)
    return dict(metric="simplex pivots/sec at LP shapes other than the headline's", unit="pivots/s", shapes=rows,
                sample="per shape: (t[K=768] - t[K=256]) / 512 on gen.hard_lp_f64(m, n), better of two passes after a warm-up pass; "
                       "pass = HIP events on the full-batch sweep launches of the last pass; chain_us_per_stage = (time per batch - pass) / pivots per pass")


def leg_six_e2e(ctx, xpoly_amd, gen, m=4096, n=8192, max_iter=1280, with_reference=True):
    """What a caller of SIX::maxm / minm waits at BASELINE configs[1]: ONE xpg_six_maxm_f64 / xpg_six_minm_f64 call with host
    arrays in (leq 4096 x 8193, vc 8192 x 8193, tgtf) and status / v / sol out, max_iter = 1280 -- split into host reshaping,
    handle + upload, the device's stage 1 + pivot loop, read-back and release (xpg_six_last_profile) -- beside the raw
    host-to-device copy of the same leq array (pageable, as the caller's is) and the real reference's time for the same call
    up to its first pivot (SIX::normalize + slack + stage 1, lpsol.h:1290-1433: the 15-29 s of SURVEY section 7)."""
    import torch
    from xpoly_amd.six import six_last_profile
    out = dict(workload="gen.dense_lp_f64(%d, %d) (SURVEY 8d cfg 2b: A, c ~ U(0.1,1), b = n U(0.5,1)), vc = -I, max_iter = %d" % (m, n, max_iter))
    leq, tgtf = gen.dense_lp_f64(m, n)
    vc = np.zeros((n, n + 1)); vc[np.arange(n), np.arange(n)] = -1.0
    six = xpoly_amd.SIX(ctx, xpoly_amd.F64)
    six.set_param(0, max_iter)
    sl, st_ = gen.dense_lp_f64(64, 96)                       # warm the context and both routes' kernels
    svc = np.zeros((96, 97)); svc[np.arange(96), np.arange(96)] = -1.0
    six.maxm(st_, svc, None, sl); six.minm(st_, svc, None, sl)
    dev = torch.device("cuda", torch.cuda.current_device())
    raws = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = torch.from_numpy(leq).to(dev)
        torch.cuda.synchronize()
        raws.append((time.perf_counter() - t0) * 1e3)
        del d
    raw_ms = min(raws[1:])
    out["raw_h2d_of_leq_ms"] = round(raw_ms, 2)
    out["leq_bytes"] = int(leq.nbytes)
    out["raw_h2d_gbs"] = round(leq.nbytes / raw_ms / 1e6, 1)
    for name, fn in (("maxm", six.maxm), ("minm", six.minm)):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            st, v, sol = fn(tgtf, vc, None, leq)
            dt = (time.perf_counter() - t0) * 1e3
            pf = six_last_profile()
            if best is None or dt < best["call_ms"]:
                best = dict(call_ms=round(dt, 2), status=int(st), v=float(v), profile=pf)
        over = best["call_ms"] - best["profile"]["device_solve_ms"]
        best["overhead_over_device_solve_ms"] = round(over, 2)
        best["overhead_over_raw_h2d"] = round(over / raw_ms, 3)
        out[name] = best
    if with_reference:
        from oracle.checker import F64, Ref
        if Ref.available():
            t0 = time.perf_counter()
            Ref().two_stage(F64, leq, tgtf, 1)
            out["reference_setup_s"] = round(time.perf_counter() - t0, 2)
            out["reference_setup_note"] = ("xcom::SIX<FloatMat,Float>::TwoStageMethod(max_iter = 1) on the same LP on one host core: slack form, "
                                           "stage 1 and ONE pivot (oracle/_ref, the real reference); baseline only")
    out["note"] = ("overhead = the call's wall time minus the device's stage 1 + pivot loop; the target is <= 1.3 x the raw pageable "
                   "host-to-device copy of leq (the bytes that must cross the link once)")
    # ---- calls that END (the dense recipe above has no reachable end under the reference: it stops at max_iter): ONE xpg_six_maxm_f64
    # call, no iteration limit, on the two full-size LPs whose ends are pinned to the real reference (tests/golden/g12_end_states.json)
    del leq, vc
    gold = golden("g12_end_states.json")
    ends = []
    six.set_param(0, 0xFFFFFFFF)
    for key, mk in (("succ_big_six_max", lambda: gen.block_lp_f64(gold["succ_big_block_seeds"], wide=True)),
                    ("bench_six_max", lambda: gen.hard_lp_f64(M, NVARS))):
        rec = gold.get(key)
        if rec is None:
            continue
        l2, t2 = mk()
        n2 = l2.shape[1] - 1
        v2 = np.zeros((n2, n2 + 1)); v2[np.arange(n2), np.arange(n2)] = -1.0
        t0 = time.perf_counter()
        st, v, sol = six.maxm(t2, v2, None, l2)
        dt = (time.perf_counter() - t0) * 1e3
        pf = six_last_profile()
        ok = int(st) == rec["status"] and float(v).hex() == rec["v"] and pf["pivots"] == rec["pivots"]
        if ok and rec["status"] == 0:
            ok = checksum(sol) == rec["sol"]
        if not ok:
            sys.exit("bench.py self-check FAILED: six_e2e natural end %s: status %d v %s pivots %d, the real reference: %d %s %d"
                     % (key, st, float(v).hex(), pf["pivots"], rec["status"], rec["v"], rec["pivots"]))
        ends.append(dict(lp=rec["generator"], leq=list(l2.shape), status=int(st), v=float(v), pivots=pf["pivots"], call_ms=round(dt, 2),
                         device_solve_ms=pf["device_solve_ms"],
                         check="status, optimum bits, pivot count%s = SIX::maxm of the real reference at its natural end (tests/golden/g12_end_states.json %s)"
                               % (", solution checksum" if rec["status"] == 0 else "", key)))
        del l2, v2
    out["natural_ends"] = ends
    return out


def leg_one_call(ctx, xpoly_amd, gen, with_reference=True):
    """The reference's OWN call pattern: one tiny problem per call (Lineq::has_solution -> MIP / SIX::maxm,
    src/com/linsys.cpp:860-904; 43 us per 14 x 6 rational LP on one core). Latency of ONE call through
    the C ABI (ctypes: prebuilt argument objects, the raw foreign call timed) and through the C++ adapter classes of
    include/xpoly_amd/*.hpp on the reference's RMat objects (oracle/_ref/dropin_demo --one-call, which also times the real
    reference's classes on the same objects on one host core) -- and the batch size at which one batched call on the GPU
    overtakes a loop of reference calls. A drop-in swap of the class WITHOUT batching is a slow-down for such problems: a
    kernel launch plus a synchronisation costs more than the reference's whole solve; INTEGRATION.md section 5 has the table."""
    import ctypes as C
    from xpoly_amd._capi import lib, vp
    from xpoly_amd.six import as_kind, empty_kind, RAT
    L = lib()
    rows_out = []

    def timed(fn, budget_s=0.25, min_reps=5):
        fn()
        n, t0 = 0, time.perf_counter()
        t1 = t0
        while n < min_reps or t1 - t0 < budget_s:
            fn(); n += 1; t1 = time.perf_counter()
        return (t1 - t0) / n * 1e6

    refl = None
    if with_reference:
        from oracle.checker import Ref
        if Ref.available():
            refl = Ref()
    # ---- SIX<RMat,Rational>::maxm, 14 x 6 and 28 x 12 (integer data, x >= 0)
    for m, nv in ((14, 6), (28, 12)):
        leq, tg = gen.int_lp_rat(m, nv, seed=gen.XS_SEED + m)
        vc = gen.to_rat(gen.vc_nonneg(nv, False))
        v = empty_kind((1,), RAT); sol = empty_kind((nv + 1,), RAT)
        args = (ctx._h, vp(tg), vp(vc), C.c_int(nv), None, C.c_int(0), vp(leq), C.c_int(m), C.c_int(nv + 1), C.c_uint(0xFFFFFFFF), vp(v), vp(sol))
        st = L.xpg_six_maxm_rat32(*args)
        rec = dict(call="SIX<RMat,Rational>::maxm", shape="%dx%d" % (m, nv), status=int(st), c_abi_us=round(timed(lambda: L.xpg_six_maxm_rat32(*args)), 1))
        # the batch entry point on nb copies: where does one GPU call overtake nb reference calls?
        per_lp = {}
        for nb in (1, 16, 256, 4096):
            bl = np.ascontiguousarray(np.broadcast_to(leq, (nb,) + leq.shape)); bt = np.ascontiguousarray(np.broadcast_to(tg, (nb,) + tg.shape))
            per_lp[nb] = round(timed(lambda: ctx.six_batch(RAT, True, bt, bl), 0.15, 3) / nb, 3)
        rec["batched_us_per_lp"] = per_lp
        if refl is not None:
            want = refl.six_solve(RAT, True, tg, vc, None, leq)
            assert want[0] == st and want[1].tolist() == v[0].tolist(), "one_call: result differs from the reference"
            rec["reference_us"] = round(timed(lambda: refl.six_solve(RAT, True, tg, vc, None, leq)), 1)
            rec["break_even_batch"] = next((nb for nb in sorted(per_lp) if per_lp[nb] < rec["reference_us"]), None)
        rows_out.append(rec)
    # ---- Lineq::has_solution(integer, unique) on a 12 x 5 dependence polyhedron
    rng = np.random.default_rng(5)
    sysm = gen.random_system(rng, 12, 4); sysm[..., 1] = 1
    vc = gen.to_rat(gen.vc_nonneg(4, False))
    a = as_kind(sysm, RAT, 2)
    hargs = (ctx._h, vp(a), C.c_int(12), None, C.c_int(0), vp(vc), C.c_int(4), C.c_int(5), C.c_int(4), C.c_int(1), C.c_int(1))
    r = L.xpg_has_solution_rat32(*hargs)
    rec = dict(call="Lineq::has_solution(int, unique)", shape="12x5", status=int(r), c_abi_us=round(timed(lambda: L.xpg_has_solution_rat32(*hargs)), 1))
    from xpoly_amd.six import dep_is_empty_batch
    per = {}
    for nb in (1, 16, 256, 4096):
        stack = np.ascontiguousarray(np.broadcast_to(a, (nb,) + a.shape))
        per[nb] = round(timed(lambda: dep_is_empty_batch(ctx, stack), 0.15, 3) / nb, 3)
    rec["batched_us_per_polyhedron (dep_is_empty: reduce + has_solution)"] = per
    if refl is not None:
        rec["reference_us"] = round(timed(lambda: refl.has_solution(sysm, None, vc, 4, True, True)), 1)
        rec["break_even_batch"] = next((nb for nb in sorted(per) if per[nb] < rec["reference_us"]), None)
    rows_out.append(rec)
    # ---- MIP<RMat,Rational>::maxm(is_bin), 24-variable knapsack with two capacity rows
    kl, kt = gen.knapsack_batch_rat(1, 24)
    kvc = gen.to_rat(gen.vc_nonneg(24, False))
    v = empty_kind((1,), RAT); sol = empty_kind((25,), RAT)
    k_tg, k_leq = np.ascontiguousarray(kt[0]), np.ascontiguousarray(kl[0])
    margs = (ctx._h, vp(k_tg), vp(kvc), C.c_int(24), None, C.c_int(0), vp(k_leq), C.c_int(k_leq.shape[0]), C.c_int(25), C.c_int(1), None, vp(v), vp(sol))
    st = L.xpg_mip_maxm_rat32(*margs)
    rec = dict(call="MIP<RMat,Rational>::maxm(is_bin)", shape="26x25", status=int(st), c_abi_us=round(timed(lambda: L.xpg_mip_maxm_rat32(*margs), 0.5, 3), 1))
    per = {}
    for nb in (1, 16, 256):
        bl, bt = gen.knapsack_batch_rat(nb, 24)
        per[nb] = round(timed(lambda: xpoly_amd.six.mip_batch(ctx, True, True, bt, bl), 0.3, 2) / nb, 3)
    rec["batched_us_per_mip"] = per
    if refl is not None:
        want = refl.mip_solve(RAT, True, True, kt[0], kvc, None, kl[0])
        assert want[0] == st and want[1].tolist() == v[0].tolist(), "one_call: MIP result differs from the reference"
        rec["reference_us"] = round(timed(lambda: refl.mip_solve(RAT, True, True, kt[0], kvc, None, kl[0]), 0.5, 2), 1)
        rec["break_even_batch"] = next((nb for nb in sorted(per) if per[nb] < rec["reference_us"]), None)
    rows_out.append(rec)
    out = dict(metric="latency of ONE call at the reference's own call pattern", unit="us", calls=rows_out,
               sample="mean over >= 0.25 s of back-to-back calls after a warm-up call; c_abi_us = the foreign call alone (ctypes, argument objects "
                      "prebuilt); reference_us = the real reference (oracle/_ref via ref_driver.cpp) on ONE host core, incl. loading its matrices; "
                      "batched_* = one batch call on nb problems / nb; break_even_batch = the smallest measured nb whose per-problem time beats the reference")
    demo = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")
    if with_reference and os.path.exists(demo):
        r = subprocess.run([demo, "--one-call"], capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            try:
                out["cxx_adapter"] = dict(json.loads(r.stdout.strip().splitlines()[-1]),
                                          note="oracle/_ref/dropin_demo --one-call: xcom::SIX / MIP / Lineq (the real reference, one core) and xpoly_amd::SIX / MIP / "
                                               "Lineq (include/xpoly_amd/*.hpp -> C ABI -> GPU) on the SAME RMat objects, object construction included; us per call")
            except ValueError:
                out["cxx_adapter"] = dict(error=r.stdout[-300:])
        else:
            out["cxx_adapter"] = dict(error=(r.stderr or r.stdout)[-300:])
    return out


RAT_M, RAT_N, RAT_K = 1024, 1023, 16                  # tableau 1024 x (1023 + 1024 + 1) = 1024 x 2048


def leg_rational(ctx, xpoly_amd, gen):
    """BASELINE configs[3]: exact rational simplex (int32 num/den, int64 intermediates, the reference's
    float32 `appro` rescue, src/com/rational.cpp:163-226), tableau 1024 x 2048, K = 16 pivots from the
    slack form (crosses the first appro activations). Integer-ALU bound (Euclid loops), not HBM."""
    RUNNING = xpoly_amd.six.XPG_RUNNING
    leq, tgtf = gen.int_lp_rat(RAT_M, RAT_N)
    lp = xpoly_amd.DeviceLP(ctx, xpoly_amd.RAT, leq, tgtf)
    reps, best = 6, None
    for rep in range(reps):
        lp.begin()
        ctx.sync()
        t0 = time.perf_counter()
        st = lp.iterate(RAT_K)
        dt = time.perf_counter() - t0
        assert st == RUNNING
        if rep > 0:
            best = dt if best is None else min(best, dt)
    # self-check, outside the timing: the state the last run left (K = 16) against the real reference's fixture
    rec = [r for r in golden("g8_large.json")["g4_large"] if r["K"] == RAT_K][0]
    got = lp.read()
    if (checksum(got["tab"]) != rec["tab"] or checksum(got["tgtf"]) != rec["tgtf"] or
            checksum(got["eq2bv"].astype(np.int32)) != rec["eq2bv"] or got["tgtf"][got["rhs"]].tolist() != rec["obj_const"]):
        sys.exit("bench.py self-check FAILED: rational tableau after %d pivots differs from tests/golden/g8_large.json (g4_large)" % RAT_K)
    del got
    lp.close()
    W = RAT_N + RAT_M + 1
    alg = 2 * RAT_M * W * 8
    issue = None
    if os.path.exists(PMC_RATIONAL):                     # committed counter summary of the same leg (tools/lab/run_rat_pmc.sh)
        k = json.load(open(PMC_RATIONAL))["kernels"].get("k_pipe_fused_r32")
        if k:
            issue = dict(kernel="k_pipe_fused_r32", valu_wave_instructions_per_launch=round(k["SQ_INSTS_VALU"]),
                         active_lanes_per_valu_instruction=k.get("active_lanes_per_valu_instruction"),
                         valu_active_us_at_2p4ghz=round(k["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2400.0, 2),
                         source="profiles/round4_pmc_rational_issue.json (rocprofv3 --pmc SQ_* on bench.py --legs rational)")
    return dict(metric="exact rational simplex pivots/sec (tableau 1024x2048, K=16)", value=round(RAT_K / best, 1),
                unit="pivots/s", us_per_pivot=round(best / RAT_K * 1e6, 2), tableau=[RAT_M, W], dtype="int32 num/den",
                bound="integer issue (gcd / appro per cell) in the late pivots, the pick -> staging latency chain inside the launch in the early ones (fused loop: one launch per pivot sweeps, picks the next pivot and stages it); not HBM",
                algorithmic_bytes_per_pivot=alg, achieved_gbs=round(alg * RAT_K / best / 1e9, 1),
                hbm_frac=round(alg * RAT_K / best / 1e9 / HBM_PEAK_GBS, 4), issue_rate=issue,
                self_check="tableau 1024x2048 (CRC-32 + sum + xor of all (num, den) cells), objective row, basis after the last run's 16 pivots = "
                           "the real reference's (tests/golden/g8_large.json g4_large, %d appro calls on the way): bit-identical" % rec["appro_calls"],
                sample="best of %d runs of xpg_lp_begin + xpg_lp_iterate(16)" % (reps - 1))


def cpu_rational(port, gen):
    from oracle.checker import RAT
    leq, tgtf = gen.int_lp_rat(RAT_M, RAT_N)
    ts = {}
    for K in (1, 3):
        t0 = time.perf_counter()
        port.two_stage(RAT, leq, tgtf, K)
        ts[K] = time.perf_counter() - t0
    per = (ts[3] - ts[1]) / 2.0
    return dict(value=round(1.0 / per, 3), unit="pivots/s", cores=1, kind="port",
                sample="oracle TwoStageMethod on the cfg-4 LP, (t[K=3]-t[K=1])/2 = %.3f s/pivot" % per)


MIP_NB, MIP_NV = 1024, 24


def leg_mip(ctx, xpoly_amd, gen):
    """BASELINE configs[4]: 0-1 knapsacks, MIP::maxm(is_bin) (src/com/lpsol.h:2427-2612). The reference's
    depth-first order decides the result, so a tree is not split: MIP_NB independent trees, one workgroup each,
    the whole tree walk on the device (mip_kernels.hip.h: node rebuild, normalisation, LDS solve, recursion)."""
    from xpoly_amd.six import mip_batch
    leq, tgtf = gen.knapsack_batch_rat(MIP_NB, MIP_NV)
    g = golden("g10_mip_bench.json")                    # the 128 knapsacks of the reference fixture lead the timed batch
    assert g["nv"] == MIP_NV
    g_leq, g_tg = gen.knapsack_batch_rat(g["nb"], g["nv"])
    leq[:g["nb"]] = g_leq; tgtf[:g["nb"]] = g_tg
    mip_batch(ctx, True, True, tgtf, leq)               # warm (also sizes the handle's device scratch)
    t0 = time.perf_counter()
    st, v, sol, nodes = mip_batch(ctx, True, True, tgtf, leq)
    dt = time.perf_counter() - t0
    for b, want in enumerate(g["results"]):             # self-check, outside the timing
        if want is None:
            ok = st[b] == -7                            # the reference is undefined there; the ABI says so
        else:
            ok = st[b] == want["status"] and [int(v[b][0]), int(v[b][1])] == want["v"]
            if ok and want["status"] == 0:
                ok = [int(x) for x in sol[b].reshape(-1)] == want["sol"]
        if not ok:
            sys.exit("bench.py self-check FAILED: MIP %d of the timed batch differs from tests/golden/g10_mip_bench.json" % b)
    big = 8 * MIP_NB                                    # the same call with 8x the trees: throughput, not tree depth
    leq8, tgtf8 = gen.knapsack_batch_rat(big, MIP_NV)
    mip_batch(ctx, True, True, tgtf8, leq8)
    t0 = time.perf_counter()
    _, _, _, nodes8 = mip_batch(ctx, True, True, tgtf8, leq8)
    dt8 = time.perf_counter() - t0
    # the opt-in NON-parity mode beside it: the same 1024 / 8192 knapsacks as fp64 programs through the
    # warm-started branch and bound, batch form -- one tree per workgroup, dual simplex from the parent's tableau in LDS
    from xpoly_amd.six import mip_warm_batch
    wl, wt = leq[..., 0].astype(np.float64), tgtf[..., 0].astype(np.float64)
    mip_warm_batch(ctx, True, wt, wl, is_bin=True)
    t0 = time.perf_counter()
    wst, wv, _, wstats = mip_warm_batch(ctx, True, wt, wl, is_bin=True)
    wdt = time.perf_counter() - t0
    solved = st == 0                                     # where the parity walk finds the optimum too the values must agree
    # (the two legs answer different questions: on 0-1 programs the reference's walk substitutes the branch equalities with
    # lpsol.h:1232's row index and returns points that violate the capacity rows on most of these knapsacks -- "optima" ABOVE
    # the optimum, reproduced bit for bit by the parity leg; how many of them the true optimum happens to equal is on file)
    pv = v[solved][:, 0] / np.maximum(v[solved][:, 1], 1)
    A_, b_ = wl[solved][:, :, :MIP_NV], wl[solved][:, :, MIP_NV]
    px = sol[solved][:, :MIP_NV, 0] / np.maximum(sol[solved][:, :MIP_NV, 1], 1)
    parity_feasible = round(float(np.mean((np.einsum("bij,bj->bi", A_, px) <= b_ + 1e-9).all(axis=1))), 3) if solved.any() else None
    equal_share = round(float(np.mean(np.abs(wv[solved] - pv) <= 1e-6)), 3) if solved.any() else None
    wl8, wt8 = leq8[..., 0].astype(np.float64), tgtf8[..., 0].astype(np.float64)
    mip_warm_batch(ctx, True, wt8, wl8, is_bin=True)
    t0 = time.perf_counter()
    _, _, _, wstats8 = mip_warm_batch(ctx, True, wt8, wl8, is_bin=True)
    wdt8 = time.perf_counter() - t0
    warm = dict(mode="OPT-IN, NON-PARITY: xpg_mip_warm_batch_f64 (best incumbent, bounding by the relaxation, dual simplex warm-started from the "
                     "parent's tableau; checked against scipy / HiGHS in tests/test_gpu_warm_mip.py, not against the reference's walk)",
                problems=MIP_NB, wall_ms=round(wdt * 1e3, 2), mips_per_s=round(MIP_NB / wdt, 1), nodes=int(wstats["nodes"]),
                nodes_per_s=round(wstats["nodes"] / wdt, 1), dual_pivots_per_node=round(wstats["dual_pivots"] / max(1, wstats["nodes"] - MIP_NB), 2),
                root_pivots_per_problem=round(wstats["root_pivots"] / MIP_NB, 1), max_depth=int(wstats["max_depth"]),
                status_hist=np.bincount(np.clip(wst, 0, 4), minlength=5).tolist(),
                share_of_the_parity_walks_points_that_satisfy_their_rows=parity_feasible, share_of_parity_answers_equal_to_the_optimum=equal_share,
                larger_batch=dict(problems=big, mips_per_s=round(big / wdt8, 1), nodes_per_s=round(wstats8["nodes"] / wdt8, 1), wall_ms=round(wdt8 * 1e3, 2)))
    return dict(metric="0-1 MIP branch and bound, one tree per workgroup on the device", value=round(nodes / dt, 1), unit="nodes/s",
                warm_started_batch=warm,
                mips_per_s=round(MIP_NB / dt, 1), problems=MIP_NB, vars=MIP_NV, rows=2 + MIP_NV, nodes=int(nodes),
                nodes_per_problem=round(nodes / MIP_NB, 2), wall_ms=round(dt * 1e3, 2),
                status_hist=np.bincount(np.clip(st, 0, 4), minlength=5).tolist(), dtype="int32 num/den",
                larger_batch=dict(problems=big, mips_per_s=round(big / dt8, 1), nodes_per_s=round(nodes8 / dt8, 1),
                                  wall_ms=round(dt8 * 1e3, 2)),
                self_check="128 MIPs of the timed batch (tests/golden/g10_mip_bench.json: the real reference's MIP<RMat,Rational>::maxm(is_bin)): "
                           "status, value and solution identical (4 on which the reference is undefined report XPG_ERR_REF_UNDEFINED)",
                sample="xpg_mip_batch_rat32, host arrays in and out (PCIe included); a batch takes as long as its deepest tree")


def leg_lineq(ctx, xpoly_amd, gen):
    """The small exact problems either side of the LP: batches of rational systems through
    Lineq::reduce / fme / Matrix::rank (src/com/linsys.cpp:359-626, :656-774, matt.h:2614-2726) at the dependence
    tests' shapes -- resident in HBM through the *_dev entry points, and through the host-array entry points (PCIe
    and the cap-row result slots included) -- and batches of small rational LPs and of dependence polyhedra through
    SIX / MIP. Integer-issue bound (gcd loops)."""
    import torch
    from xpoly_amd import lineq as LQ
    from xpoly_amd.lineq import Lineq
    lq = Lineq(ctx)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    rows_out = []
    for rows, nv in ((16, 8), (40, 12), (60, 19)):
        cols = nv + 1
        base = np.stack([gen.random_system(rng, rows, nv) for _ in range(256)])
        mats = np.ascontiguousarray(np.tile(base, (LINEQ_NB // 256, 1, 1, 1)))
        rec = dict(rows=rows, cols=cols)
        # (i) systems resident in HBM, results left there: the *_dev entry points, timed around xpg_sync
        cap = max(rows, rows * rows // 4 + rows + 1)
        d_in = torch.from_numpy(mats).to(dev)
        d_work = torch.empty_like(d_in)
        d_out = torch.zeros(LINEQ_NB, cap, cols, 2, dtype=torch.int32, device=dev)
        d_r = torch.empty(LINEQ_NB, dtype=torch.int32, device=dev); d_k = torch.empty_like(d_r)

        def resident(name):
            if name == "reduce":
                d_work.copy_(d_in); torch.cuda.synchronize()
                t0 = time.perf_counter()
                LQ.reduce_dev(ctx, LINEQ_NB, d_work.data_ptr(), rows, cols, nv, True, d_r.data_ptr(), d_k.data_ptr())
            elif name == "fme":
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                LQ.fme_dev(ctx, LINEQ_NB, d_in.data_ptr(), rows, cols, nv, 0, False, d_out.data_ptr(), cap, d_r.data_ptr(), d_k.data_ptr())
            else:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                LQ.rank_dev(ctx, LINEQ_NB, d_in.data_ptr(), rows, cols, d_r.data_ptr())
            ctx.sync()
            return time.perf_counter() - t0
        for name in ("reduce", "fme", "rank"):
            resident(name)
            rec[name + "_systems_per_s"] = round(LINEQ_NB / min(resident(name) for _ in range(3)), 0)
        del d_in, d_work, d_out, d_r, d_k
        # (ii) the host-array entry points: PCIe and the cap-row result slots included
        work = mats.copy()                              # (the in-place form overwrites its argument)
        for name, fn in (("reduce", lambda: lq.reduce_packed(mats, nv, True, copy=False)),
                         ("reduce_inplace", lambda: lq.reduce_inplace(work, nv, True)), ("fme_slots", lambda: lq.fme(mats, nv, 0, slots=True)),
                         ("fme", lambda: lq.fme_packed(mats, nv, 0, copy=False)), ("rank", lambda: lq.rank(mats))):
            fn()
            t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
            rec[name + "_host_arrays_systems_per_s"] = round(LINEQ_NB / dt, 0)
        # what the link allows for the packed form: the systems up, the live rows down
        _, off, _ = lq.fme_packed(mats[:256], nv, 0)
        bytes_per_system = rows * cols * 8 + float(off[-1]) / 256 * cols * 8
        rec["fme_result_rows_mean"] = round(float(off[-1]) / 256, 1)
        rec["fme_host_arrays_pcie_bound_systems_per_s"] = round(PCIE_GBS * 1e9 / bytes_per_system, 0)
        one = mats[:1]
        lq.fme_packed(one, nv, 0)
        t0 = time.perf_counter()
        for _ in range(200):
            lq.fme_packed(one, nv, 0, copy=False)
        rec["fme_one_system_call_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
        rows_out.append(rec)
    # small rational LPs, dependence-test-like integer data (12 rows, 16 variables)
    nb, m, cols = 8192, 12, 17
    leq, tg = gen.small_lp_batch_f64(nb, m, cols, 1)
    rl, rt = gen.to_rat(leq.astype(np.int32)), gen.to_rat(tg.astype(np.int32))
    ctx.six_batch(xpoly_amd.RAT, True, rt[:64], rl[:64])
    t0 = time.perf_counter(); st, _, _ = ctx.six_batch(xpoly_amd.RAT, True, rt, rl); dt = time.perf_counter() - t0
    # DepPoly::is_empty (src/eng/poly.cpp:530-573): reduce, then has_solution(integer, unique) = MIP maxm / minm
    from xpoly_amd.six import dep_is_empty_batch
    dnb, drows, dnv = 4096, 12, 4
    dm = np.stack([gen.random_system(rng, drows, dnv) for _ in range(256)])
    dm[..., 1] = 1                                     # dependence polyhedra are integer systems
    dm = np.ascontiguousarray(np.tile(dm, (dnb // 256, 1, 1, 1)))
    dep_is_empty_batch(ctx, dm[:256])
    t0 = time.perf_counter(); empty, dnodes = dep_is_empty_batch(ctx, dm); ddt = time.perf_counter() - t0
    dbig = 16 * dnb                                    # the same call with 16x the polyhedra: throughput
    dm16 = np.ascontiguousarray(np.tile(dm, (16, 1, 1, 1)))
    dep_is_empty_batch(ctx, dm16)
    t0 = time.perf_counter(); dep_is_empty_batch(ctx, dm16); ddt16 = time.perf_counter() - t0
    return dict(metric="rational row elimination and small rational LPs, batched", systems=LINEQ_NB, shapes=rows_out,
                dep_is_empty=dict(polyhedra=dnb, rows=drows, vars=dnv, polyhedra_per_s=round(dnb / ddt, 0),
                                  nodes=int(dnodes), empty=int(np.sum(empty == 1)),
                                  larger_batch=dict(polyhedra=dbig, polyhedra_per_s=round(dbig / ddt16, 0))),
                rational_lps=dict(lps=nb, rows=m, cols=cols, lps_per_s=round(nb / dt, 0),
                                  status_hist=np.bincount(np.clip(st, 0, 4), minlength=5).tolist()),
                dtype="int32 num/den", bound="integer issue (gcd loops), not HBM",
                sample="*_systems_per_s: 16384 systems resident in HBM, best of 3 calls timed to xpg_sync; *_host_arrays_*, "
                       "rational_lps and dep_is_empty: host arrays in and out (PCIe included), second call timed; reduce_host_arrays = the packed entry "
                       "point xpg_lineq_reduce_batch_packed_rat32 (survivors written by the device straight into pinned memory, one "
                       "synchronisation), reduce_inplace_host_arrays = the reference-shaped in-place form on top of it; fme_host_arrays = "
                       "the packed entry point (row offsets + live rows through pinned memory), fme_slots_host_arrays = round 2's "
                       "cap-row slots; *_pcie_bound_* = 63 GB/s / (system bytes up + mean live-row bytes down); "
                       "fme_one_system_call_us = mean of 200 one-system packed calls incl. the ctypes layer")


if __name__ == "__main__":
    main()
