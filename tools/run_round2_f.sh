cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof -- python3 bench.py --legs rational --no-cpu-baseline > gpurun_out/r2f/bench.log 2>&1
f=$(find gpurun_out/r2f/prof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-180
cp $f gpurun_out/r2f/rational_kernel_stats.csv
find gpurun_out/r2f/prof -name "*kernel_trace.csv" -delete
