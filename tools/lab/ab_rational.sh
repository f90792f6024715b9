cd $GRAFT_REPO_ROOT
for r in 8 4 2 1; do
  XPG_R32_ROWS=$r python bench.py --legs rational --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', $r, d['rational']['value'], d['rational']['us_per_pivot'])"
done
