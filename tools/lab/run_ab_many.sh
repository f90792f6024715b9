# A/B of several builds inside one GPU run: every library named in $2.. (default: all of tools/_build/libxpoly_*.so),
# bench legs in $1, twice each, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LEGS=${1:-pivots,cfg2b}; shift
SOS=${@:-$(ls tools/_build/libxpoly_*.so | grep -v stamps)}
for rep in 1 2; do for so in $SOS; do
  XPG_SO_PATH=$PWD/$so python bench.py --legs $LEGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1])
o=['$so']
if 'batched' in d: o += [(k, v['lps_per_s']) for k, v in d['batched']['families'].items()]
if d.get('value'): o += ['pivots/s', d['value']]
for leg in ('cfg2b','rational','mip'):
    if leg in d: o += [leg, d[leg].get('value')]
print(*o)
"
done; done 2>&1 | tee gpurun_out/ab_many.log
