#!/bin/bash
# A/B of two builds on chosen bench legs: run_ab_so.sh <legs> <other .so>
R=$GRAFT_REPO_ROOT
for k in 1 2; do
  for so in "" $2; do
    XPG_SO_PATH=$so python $R/bench.py --legs $1 --no-cpu-baseline --no-ref-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); out={}
if 'mip' in d and d['mip']: out['mip_ms']=d['mip']['wall_ms']; out['mip8192']=round(d['mip']['larger_batch']['mips_per_s'])
if 'batched' in d and d['batched']: out.update({f: round(v['lps_per_s']) for f, v in d['batched']['families'].items()})
if 'rational' in d and d['rational']: out['rational']=d['rational']['value']
print('${so:-default}', out)"
  done
done
