"""CPU suite, part 3: the N > 1 path (contiguous sharding + the single final gather) with
two processes over gloo. No solver runs here -- each rank fabricates the records of its own
shard from the global LP index, so the test checks exactly what the multi-GPU code adds:
slice ownership, ragged shards and the order of the gathered result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_everything():
    sys.path.insert(0, ROOT)
    from xpoly_amd.shard import shard_range
    for total in (0, 1, 7, 8, 65536, 65537, 100001):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                assert 0 <= lo <= hi <= total
                covered += list(range(lo, hi)) if total < 100 else [(lo, hi)]
            if total < 100:
                assert covered == list(range(total))
            else:
                assert covered[0][0] == 0 and covered[-1][1] == total
                assert all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
                sizes = [b - a for a, b in covered]
                assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, width, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from xpoly_amd.shard import gather_records, pack_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    status = (idx % 5).to(torch.int32)
    v = idx * 0.5
    sol = idx[:, None] + torch.arange(width, dtype=torch.float64)[None, :] / 100.0
    rec = pack_records(status, v, sol)
    full = gather_records(rec, total, rank, world, dist)
    dist.barrier()
    q.put((rank, full.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 11])
def test_gather_two_ranks_gloo(total):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    width = 6
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, width, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(total, dtype=np.float64)
    want = np.concatenate([(idx % 5)[:, None], (idx * 0.5)[:, None],
                           idx[:, None] + np.arange(width)[None, :] / 100.0], axis=1)
    for r in (0, 1):
        assert got[r].shape == (total, 2 + width)
        assert np.array_equal(got[r], want)


def _worker_rat(rank, world, port, total, cols, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from xpoly_amd.shard import gather_records, pack_records_i32, pack_records_rat, shard_range, unpack_records_rat
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    idx = np.arange(lo, hi, dtype=np.int64)
    st = (idx % 4).astype(np.int32)
    v = np.stack([2147483647 - idx, idx + 1], axis=1).astype(np.int32)          # values no float64 column of a mixed record would keep apart from status
    sol = (idx[:, None, None] * 3 + np.arange(cols)[None, :, None] * 2 + np.arange(2)[None, None, :]).astype(np.int32)
    full = gather_records(pack_records_rat(st, v, sol), total, rank, world, dist)
    verd = gather_records(pack_records_i32((idx % 3).astype(np.int32), (idx * 7).astype(np.int32)), total, rank, world, dist)
    gst, gv, gsol = unpack_records_rat(full)
    dist.barrier()
    q.put((rank, gst.numpy().copy(), gv.numpy().copy(), gsol.numpy().copy(), verd.numpy().copy(), str(full.dtype)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [9, 64])
def test_gather_exact_rational_records_two_ranks_gloo(total):
    """The rational legs (MIP trees, dependence verdicts) gather int32 records: (status, v num/den, sol num/den) come back
    in global problem order, bit for bit, on every rank; shards ragged by one."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cols = 5
    procs = [ctx.Process(target=_worker_rat, args=(r, 2, port, total, cols, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(total, dtype=np.int64)
    for rank, st, v, sol, verd, dtype in got:
        assert dtype == "torch.int32"
        assert np.array_equal(st, idx % 4)
        assert np.array_equal(v, np.stack([2147483647 - idx, idx + 1], axis=1))
        assert np.array_equal(sol, idx[:, None, None] * 3 + np.arange(cols)[None, :, None] * 2 + np.arange(2)[None, None, :])
        assert np.array_equal(verd, np.stack([idx % 3, idx * 7], axis=1))


def test_bench_spawns_its_own_ranks_and_gathers():
    """`python bench.py --gpus 2` starts two ranks by itself (torch.distributed.run on 127.0.0.1) and the
    batched leg's shard -> solve -> all_gather path runs in them; here with --backend gloo and a stub
    solver (each rank fabricates the records of its own shard), so exactly bench.py's own spawn and
    gather code is what is exercised. The gathered order is asserted inside bench.py."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--stub-solver", "--legs", "batched,sharded", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["stub_solver"] is True
    b = out["batched"]
    assert b["ranks"] == 2 and b["total_lps"] == 2 * 8192 and b["lps_per_rank"] == [8192, 8192]
    assert "world size 2" in b["collective"]
    assert b["families"]["dep_test_like"]["gather_ms"] > 0
    # the exact (int32-record) legs of BASELINE configs[4]: MIP trees and dependence verdicts, sharded the same way;
    # the gathered order and every field are asserted inside bench.py (leg_sharded)
    sh = out["sharded"]
    assert sh["ranks"] == 2 and sh["mip"]["problems_total"] == 2 * 1024 and sh["dep_is_empty"]["problems_total"] == 2 * 4096
    assert sh["mip"]["record_int32s"] == 3 + 2 * 25 and "world size 2" in sh["collective"]


def test_bench_without_gpus_fails_loudly():
    """On a box without GPUs `bench.py --gpus 2` must not degrade to a silent 1-rank run: every rank
    fails creating its context (XPG_ERR_NO_DEVICE) and the parent exits non-zero."""
    import subprocess
    from xpoly_amd import _capi
    if _capi.lib().xpg_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--legs", "batched",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "XPG_ERR_NO_DEVICE" in r.stderr or "No HIP GPUs" in r.stderr or "no GPU" in r.stderr.lower(), r.stderr[-3000:]


def test_numa_placement_reads_sysfs_only(tmp_path):
    """A rank binds itself to the NUMA node of its GPU before its first GPU call (bench.py, xpoly_amd/shard.py): the node
    comes from sysfs -- AMD display / accelerator functions in bus order -- and an unknown topology changes nothing."""
    import os
    from xpoly_amd.shard import _parse_cpulist, gpu_numa_nodes, pin_to_gpu_numa
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    devs = tmp_path / "bus" / "pci" / "devices"
    for bdf, vendor, cls, node in (("0000:05:00.0", "0x1002", "0x120000", "0"), ("0000:25:00.0", "0x1002", "0x120000", "1"),
                                   ("0000:01:00.0", "0x8086", "0x020000", "0"), ("0000:45:00.0", "0x1002", "0x030000", "-1")):
        d = devs / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n"); (d / "class").write_text(cls + "\n"); (d / "numa_node").write_text(node + "\n")
    me = sorted(os.sched_getaffinity(0))
    for n, cl in ((0, "%d" % me[0]), (1, "%d" % me[-1])):
        nd = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    assert gpu_numa_nodes(str(tmp_path)) == [("0000:05:00.0", 0), ("0000:25:00.0", 1), ("0000:45:00.0", -1)]
    r = pin_to_gpu_numa(1, sysfs=str(tmp_path), apply=False)
    assert r == {"gpu": "0000:25:00.0", "node": 1, "cpus": 1, "pinned": False}
    assert pin_to_gpu_numa(2, sysfs=str(tmp_path), apply=False)["node"] == -1          # no node reported: left alone
    assert pin_to_gpu_numa(7, sysfs=str(tmp_path), apply=False)["gpu"] is None           # no such GPU
    assert pin_to_gpu_numa(0, sysfs=str(tmp_path / "nothing"), apply=False)["gpu"] is None
    assert sorted(os.sched_getaffinity(0)) == me
