#!/bin/bash
R=$GRAFT_REPO_ROOT
for k in 1; do
for s1 in 1024 2048 3072 4096 8192; do
  XPG_BATCH_SLICE_STAGE1=$s1 python $R/bench.py --legs batched --no-cpu-baseline --no-ref-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batched']
print('stage1 slice $s1', {f: (round(v['lps_per_s']), v.get('ms_per_pass')) for f, v in b['families'].items()})"
done
done
