#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in alone load; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chain_load_$m -- python3 $R/tools/lab/probe_chain_under_load.py $m 2>&1 | grep "pivots in"
  f=$(find $R/gpurun_out/chain_load_$m -name "*kernel_stats.csv" | head -1)
  grep "k_blk_chain\|k_blk_sweep_full\|elementwise\|copy" $f | cut -d, -f1-4,6,7 | cut -c1-160
  find $R/gpurun_out/chain_load_$m -name "*kernel_trace.csv" -delete
done
