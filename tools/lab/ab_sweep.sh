# A/B of the blocked sweep's tile order / row alignment inside the product loop (bench legs pivots + cfg2b)
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=${XPG_SO_PATH:-$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so}   # hook-only knobs: the -DXPG_TEST_HOOKS build
run() { echo "== $*"; env "$@" python bench.py --legs pivots,cfg2b --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; c=d['cfg2b']
print('  headline %.0f pivots/s  sweep %.2f us frac %.3f | cfg2b %.0f pivots/s  sweep %s us' % (d['value'], r['avg_launch_us'], r['frac'], c['value'], c.get('roofline',{}).get('avg_launch_us')))
"; }
run A=1
run XPG_SERPENTINE=0
run XPG_LD_ALIGN=64
run XPG_LD_ALIGN=32
