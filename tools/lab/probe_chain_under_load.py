"""GPU probe: how long does the blocked loop's persistent chain launch (k_blk_chain, 24 stages) take while ANOTHER stream
saturates HBM -- the question behind DESIGN 7.1 (the chain of batch b + 1 inside the launch that sweeps batch b).
Run under rocprofv3 --kernel-trace --stats twice: `alone` and `load` (a torch stream copying 1 GiB tensors back to back
while the LP iterates); compare the average duration of k_blk_chain (and of the sweep) in the two kernel_stats.csv."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import xpoly_amd
from tools import gen

mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
ctx = xpoly_amd.Context(0)
leq, tg = gen.hard_lp_f64(4096, 4095)
lp = xpoly_amd.DeviceLP(ctx, 0, leq, tg)
lp.begin()
lp.iterate(240)
side = torch.cuda.Stream()
a = torch.empty(1 << 27, dtype=torch.float64, device="cuda")      # 1 GiB
b = torch.empty_like(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
if mode == "load":
    with torch.cuda.stream(side):
        for _ in range(400):                                       # ~0.3 ms each at full rate: covers the LP's 1920 pivots
            b.copy_(a, non_blocking=True)
st = lp.iterate(1920)
dt = time.perf_counter() - t0
torch.cuda.synchronize()
print(mode, "1920 pivots in %.2f ms = %.2f us per pivot" % (dt * 1e3, dt / 1920 * 1e6))
