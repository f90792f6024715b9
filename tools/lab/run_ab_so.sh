# A/B of two builds inside one GPU run: the library in the tree against tools/_build/libxpoly_prev.so (the previous commit's
# source, built in the authoring container), bench legs in $1 (default: batched), each twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LEGS=${1:-batched}
for rep in 1 2; do for so in xpoly_amd/libxpoly_amd.so tools/_build/libxpoly_prev.so; do
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
done; done 2>&1 | tee gpurun_out/ab_so.log
