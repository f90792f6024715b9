# A/B inside one run: pivots staged per pass (XPG_BLOCK) on the two large-tableau legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for b in ${BLOCKS:-24 32 16}; do
  XPG_BLOCK=$b python bench.py --legs pivots,cfg2b --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1])
print('block $b: pivots/s', d['value'], 'chain us/stage', d['roofline']['chain']['us_per_stage'], 'sweep', d['roofline']['avg_launch_us'], '| cfg2b', d['cfg2b']['value'], 'sweep', d['cfg2b']['roofline']['avg_launch_us'], d.get('self_check',{}).get('pivots',{}).get('result'))
"
done; done 2>&1 | tee gpurun_out/block_ab.log
