#!/bin/bash
R=$GRAFT_REPO_ROOT
for k in 1 2; do
  for t in 256 128 64; do
   XPG_BATCH_THREADS=$t python $R/bench.py --legs mip --no-cpu-baseline --no-ref-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['mip']; print('threads $t', m['wall_ms'], round(m['mips_per_s']), round(m['larger_batch']['mips_per_s']))"
  done
done
