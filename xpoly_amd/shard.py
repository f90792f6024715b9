"""Sharding of independent LPs / systems across the GPUs of one node.

The polyhedral workload is thousands of independent feasibility problems per SCoP
(src/eng/poly.cpp:530-573 -> src/com/linsys.cpp:830-906); nothing is shared between
them, so each rank takes a contiguous slice, solves it with no data-path
communication, and the fixed-size result records are gathered ONCE at the end
(torch.distributed all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` items owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(status, v, sol):
    """(status, v, sol[cols]) -> one float64 record per LP: [status, v, sol...]. Works for torch
    tensors and numpy arrays; rational results are packed by the caller as two float64 columns."""
    import torch
    status = torch.as_tensor(status)
    v = torch.as_tensor(v)
    sol = torch.as_tensor(sol)
    rec = torch.empty(status.shape[0], 2 + sol.shape[1], dtype=torch.float64, device=sol.device)
    rec[:, 0] = status.to(torch.float64)
    rec[:, 1] = v
    rec[:, 2:] = sol
    return rec


def pack_records_rat(status, v, sol):
    """Exact rational results -> one int32 record per problem: [status, v.num, v.den, sol[0].num, sol[0].den, ...].
    status [n] int32, v [n, 2] int32, sol [n, cols, 2] int32 (numpy or torch; what xpg_mip_batch_rat32 /
    xpg_six_batch_rat32 return). Nothing is rounded on the way through the gather."""
    import torch
    status = torch.as_tensor(status); v = torch.as_tensor(v); sol = torch.as_tensor(sol)
    n = status.shape[0]
    rec = torch.empty(n, 3 + 2 * sol.shape[1], dtype=torch.int32, device=sol.device)
    rec[:, 0] = status.to(torch.int32)
    rec[:, 1:3] = v.reshape(n, 2).to(torch.int32)
    rec[:, 3:] = sol.reshape(n, -1).to(torch.int32)
    return rec


def unpack_records_rat(rec):
    """Inverse of pack_records_rat: (status [n], v [n, 2], sol [n, cols, 2]) as int32 tensors."""
    n = rec.shape[0]
    return rec[:, 0], rec[:, 1:3], rec[:, 3:].reshape(n, -1, 2)


def pack_records_i32(*cols):
    """Integer verdicts (e.g. DepPoly::is_empty answers, node counts) -> int32 records, one column per argument."""
    import torch
    cols = [torch.as_tensor(c).reshape(len(c), -1).to(torch.int32) for c in cols]
    return torch.cat(cols, dim=1).contiguous()


def gather_records(rec, total, rank, world, dist=None):
    """All-gathers per-rank record blocks (ragged by at most one row) into the global order.

    rec: [n_local, width] float64 tensor of this rank (n_local = shard size). Returns a
    [total, width] tensor on every rank. Without a process group (dist None) it is a no-op; with one the
    collective runs whatever the world size (bench.py --force-dist: RCCL at world size 1)."""
    import torch
    if dist is None:
        return rec
    base, extra = divmod(total, world)
    cap = base + (1 if extra else 0)
    pad = torch.zeros(cap, rec.shape[1], dtype=rec.dtype, device=rec.device)
    pad[: rec.shape[0]] = rec
    out = torch.empty(world * cap, rec.shape[1], dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r * cap: r * cap + (hi - lo)])
    return torch.cat(parts, dim=0)


# ---- host placement of a rank: the NUMA node of its GPU -------------------------------------------------------------
def _parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU function on the PCI bus, in bus order (the order HIP enumerates them in when
    HIP_VISIBLE_DEVICES does not say otherwise): [(bdf, node), ...], node -1 where the platform reports none.
    Reads sysfs only -- nothing here touches the GPU."""
    import os
    root = os.path.join(sysfs, "bus", "pci", "devices")
    out = []
    try:
        names = sorted(os.listdir(root))
    except OSError:
        return out
    for bdf in names:
        d = os.path.join(root, bdf)
        try:
            vendor = open(os.path.join(d, "vendor")).read().strip().lower()
            cls = open(os.path.join(d, "class")).read().strip().lower()
        except OSError:
            continue
        # AMD, display controller (0x03....) or processing accelerator (0x12....: how Instinct parts present themselves)
        if vendor != "0x1002" or not (cls.startswith("0x03") or cls.startswith("0x12")):
            continue
        try:
            node = int(open(os.path.join(d, "numa_node")).read().strip())
        except (OSError, ValueError):
            node = -1
        out.append((bdf, node))
    return out


def pin_to_gpu_numa(local_rank, sysfs="/sys", apply=True):
    """Binds the calling process (a rank, BEFORE its first GPU call: the pinned staging it allocates afterwards then lies on
    that node) to the CPUs of the NUMA node of GPU `local_rank`. Best effort: returns a record of what it found and did --
    {"gpu": bdf, "node": n, "cpus": k, "pinned": bool} -- and never raises; an unknown topology leaves the affinity alone."""
    import os
    rec = {"gpu": None, "node": -1, "cpus": 0, "pinned": False}
    gpus = gpu_numa_nodes(sysfs)
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    idx = local_rank
    if vis:
        try:
            idx = [int(x) for x in vis.split(",")][local_rank]
        except (ValueError, IndexError):
            return rec
    if idx < 0 or idx >= len(gpus):
        return rec
    rec["gpu"], rec["node"] = gpus[idx]
    if rec["node"] < 0:
        return rec
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % rec["node"], "cpulist")).read())
    except OSError:
        return rec
    allowed = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(cpus)
    cpus = [c for c in cpus if c in allowed]
    rec["cpus"] = len(cpus)
    if cpus and apply and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, cpus)
            rec["pinned"] = True
        except OSError:
            pass
    return rec
