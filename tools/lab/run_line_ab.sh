# A/B inside one run (boxes differ by a per cent or two): the pick workers' entering-column line in LDS (XPG_CHAIN_LINE) and the
# column-major copy of the staged rows (XPG_CHAIN_ET), both blocked legs, every setting twice
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=${XPG_SO_PATH:-$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so}   # hook-only knobs: the -DXPG_TEST_HOOKS build
mkdir -p gpurun_out
for rep in 1 2; do for mode in "0 0" "1 0"; do
  set -- $mode
  XPG_CHAIN_LINE=$1 XPG_CHAIN_ET=$2 python bench.py --legs pivots,cfg2b --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('line $1 et $2: pivots/s', d['value'], 'chain us/stage', d['roofline']['chain']['us_per_stage'], 'sweep', d['roofline']['avg_launch_us'], '| cfg2b', d['cfg2b']['value'], 'sweep', d['cfg2b']['roofline']['avg_launch_us'], d.get('self_check',{}).get('pivots',{}).get('result'))
"
done; done 2>&1 | tee gpurun_out/line_ab.log
