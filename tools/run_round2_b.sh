# round 2, GPU call B: the new GPU tests, launch lab, per-kernel trace gaps, overlap probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2b/pytest.log
tools/_build/launch_lab > gpurun_out/r2b/launch_lab.log 2>&1; cat gpurun_out/r2b/launch_lab.log
PYTHONPATH=$GRAFT_REPO_ROOT python tools/probe_overlap.py > gpurun_out/r2b/overlap.log 2>&1; cat gpurun_out/r2b/overlap.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/prof -- python3 bench.py --steps 1 --warmup 0 --legs pivots --no-cpu-baseline > gpurun_out/r2b/prof_bench.log 2>&1
f=$(find gpurun_out/r2b/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f > gpurun_out/r2b/gaps.log 2>&1; cat gpurun_out/r2b/gaps.log
rm -rf gpurun_out/r2b/prof
