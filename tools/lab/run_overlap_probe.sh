# Timing experiment (apply tools/lab/patches/overlap_probe.patch, build with XPG_BUILD_FLAGS=-DXPG_LAB_OVERLAP into tools/_build/libxpoly_overlap.so): how long do a chain launch and a full-batch sweep take when they run NEXT to each other (a lab build,
# -DXPG_LAB_OVERLAP: the sweep of a scratch tableau on a second stream beside every chain launch)?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/overlap
for so in tools/_build/libxpoly_overlap.so; do
  n=$(basename $so .so)
  XPG_SO_PATH=$PWD/$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/overlap/$n -- python3 bench.py --legs pivots --no-cpu-baseline > gpurun_out/overlap/$n.log 2>&1
  grep -h '"metric"' gpurun_out/overlap/$n.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['value'])"
  f=$(find gpurun_out/overlap/$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -6 $f | cut -c1-200 && cp $f gpurun_out/overlap/${n}_kernel_stats.csv
  rm -rf gpurun_out/overlap/$n
done
