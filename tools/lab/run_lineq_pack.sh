#!/bin/bash
# A/B of the row-elimination kernels: one system per wave (XPG_LINEQ_LANES=64, round 1's shape) against the
# packed groups (default). Kernel times come from rocprofv3 --kernel-trace --stats.
cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=$GRAFT_REPO_ROOT TMPDIR=/tmp
mkdir -p gpurun_out/lineq
python -m pytest tests/test_gpu_lineq.py tests/test_gpu_mip.py tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/lineq/tests.log 2>&1
tail -3 gpurun_out/lineq/tests.log
for L in 64 0; do
  export XPG_LINEQ_LANES=$L
  python3 tools/lab/probe_lineq.py > gpurun_out/lineq/probe_L$L.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lineq/prof_L$L -- python3 tools/lab/probe_lineq.py > /dev/null 2>&1
  f=$(find gpurun_out/lineq/prof_L$L -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/lineq/kernel_stats_L$L.csv
  echo "== L=$L"; cat gpurun_out/lineq/probe_L$L.log; head -8 "$f"
done
