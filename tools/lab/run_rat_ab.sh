#!/bin/bash
# A/B of the rational leg: default build against tools/_build/libxpoly_$1.so (three runs each, interleaved)
R=$GRAFT_REPO_ROOT
for k in 1 2 3; do
  for so in "" tools/_build/libxpoly_$1.so; do
    XPG_SO_PATH=$so python $R/bench.py --legs rational --no-cpu-baseline --no-ref-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${so:-default}', d['rational']['value'], d['rational']['us_per_pivot'])"
  done
done
