#!/bin/bash
# A/B of k_batch's time slices on the batched leg (XPG_BATCH_SLICE: 0 = off)
R=$GRAFT_REPO_ROOT
for s in ${@:-0 512}; do
  XPG_BATCH_SLICE=$s python $R/bench.py --legs batched --no-cpu-baseline --no-ref-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batched']
print('slice $s', {f: (round(v['lps_per_s']), v.get('ms_per_pass')) for f, v in b['families'].items()}, 'self_check' in d and str(d['self_check'].get('batched'))[:80])"
done
