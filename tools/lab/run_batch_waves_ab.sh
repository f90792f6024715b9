# What a fifth LP per CU is worth to k_batch<F64> (the register-resident-cell variant needs 128 registers = four LPs per CU): the batched
# leg with XPG_BATCH_WAVES=4 against the default five, twice each, on the hooks build (the knob is hook-only)
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so
for rep in 1 2; do for w in 5 4; do
  XPG_BATCH_WAVES=$w python bench.py --legs batched --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])['batched']['families']
print('LPs per CU $w:', {k: v['lps_per_s'] for k, v in d.items()})"
done; done
