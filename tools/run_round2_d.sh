cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_large_golden.py -x -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2d/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err ) 2>&1 | grep real; echo "bench rc=$?"; cat gpurun_out/r2d/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('value','ms_per_step'): print(k, d[k])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','avg_launch_us','sweeps_in_region','pivots_per_launch','loop_effective')})
print('batched', d['batched']['value'], d['batched']['families'], d['batched'].get('n1_reference_points'))
for k in ('cfg2b','rational','mip','cpu_baseline'): print(k, d.get(k))
"; tail -5 gpurun_out/r2d/bench.err
