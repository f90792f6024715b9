# A/B: the leading dimension of the 4096 x 8192 tableau against the chain's column gathers (a power-of-two row stride puts a
# column's 4096 lines on few L2 / HBM channels) and the sweep
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=${XPG_SO_PATH:-$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so}   # hook-only knobs: the -DXPG_TEST_HOOKS build
mkdir -p gpurun_out
for pad in 0 16 32 48 64 80 144 272 528; do
  XPG_LD_PAD=$pad python bench.py --legs pivots --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('pad $pad: pivots/s', d['value'], 'sweep us', d['roofline'].get('avg_launch_us'), 'chain us/stage', d['roofline']['chain']['us_per_stage'], d.get('self_check',{}).get('pivots',{}).get('result'))
"
done 2>&1 | tee gpurun_out/ld_pad.log
