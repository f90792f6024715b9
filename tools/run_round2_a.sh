# round 2, GPU call A: parity suite, the driver's bench command, kernel stats of the pivots leg, overlap probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2a/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a/bench_driver.json 2> gpurun_out/r2a/bench_driver.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r2a/bench_driver.json
python bench.py --steps 60 --warmup 5 --legs pivots --no-cpu-baseline > gpurun_out/r2a/bench_long.json 2>&1
cut -c1-400 gpurun_out/r2a/bench_long.json
python tools/probe_overlap.py > gpurun_out/r2a/overlap.log 2>&1; cat gpurun_out/r2a/overlap.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/prof -- python3 bench.py --steps 6 --warmup 2 --legs pivots --no-cpu-baseline > gpurun_out/r2a/prof_bench.log 2>&1
f=$(find gpurun_out/r2a/prof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
find gpurun_out/r2a/prof -name "*kernel_trace.csv" -delete
