# A/B of k_fme_batch's result placement (XPG_FME_FULL_KB: layouts up to that size keep the result in LDS):
# resident fme rate per shape
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for kb in 1 6 12 20 32; do
  echo "== XPG_FME_FULL_KB=$kb"
  XPG_FME_FULL_KB=$kb python - <<'PY'
import time, numpy as np, torch, xpoly_amd
from xpoly_amd import lineq as LQ
from tools import gen
ctx = xpoly_amd.Context(0); dev = torch.device("cuda", 0); rng = np.random.default_rng(0); NB = 16384
out = []
for rows, nv in ((10, 5), (16, 8), (20, 7), (24, 8), (30, 9), (40, 12), (50, 15), (60, 19)):
    cols = nv + 1
    base = np.stack([gen.random_system(rng, rows, nv) for _ in range(256)])
    mats = np.ascontiguousarray(np.tile(base, (NB // 256, 1, 1, 1)))
    cap = max(rows, rows * rows // 4 + rows + 1)
    d_in = torch.from_numpy(mats).to(dev)
    d_out = torch.zeros(NB, cap, cols, 2, dtype=torch.int32, device=dev)
    d_r = torch.empty(NB, dtype=torch.int32, device=dev); d_k = torch.empty_like(d_r)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        LQ.fme_dev(ctx, NB, d_in.data_ptr(), rows, cols, nv, 0, False, d_out.data_ptr(), cap, d_r.data_ptr(), d_k.data_ptr()); ctx.sync()
        dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    out.append("%dx%d %.1fM" % (rows, cols, NB / best / 1e6))
    del d_in, d_out
print("  ".join(out))
PY
done
