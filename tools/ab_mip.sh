cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mip.py tests/test_gpu_multi.py -x -q 2>&1 | tail -2
python - <<'PY'
import time, numpy as np, xpoly_amd
from tools import gen
from xpoly_amd.six import mip_batch
ctx = xpoly_amd.Context(0)
for nb in (1024, 8192):
    leq, tg = gen.knapsack_batch_rat(nb, 24)
    mip_batch(ctx, True, True, tg[:64], leq[:64])
    t0 = time.perf_counter(); st, v, sol, nodes = mip_batch(ctx, True, True, tg, leq); dt = time.perf_counter() - t0
    print("nb %5d: %.1f ms, %.0f MIPs/s, %.0f nodes/s" % (nb, dt * 1e3, nb / dt, nodes / dt))
PY
