cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2c/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --legs pivots --no-cpu-baseline > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r2c/bench.json; tail -3 gpurun_out/r2c/bench.err
XPG_CHAIN=0 timeout 300 python bench.py --steps 20 --warmup 5 --legs pivots --no-cpu-baseline 2>/dev/null | cut -c1-200
