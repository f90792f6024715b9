# rocprofv3 kernel-trace A/B of the loop variants on one box: XPG_LOOP=pipe|split|serial
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in ${MODES:-serial pipe split serial pipe}; do
  rm -rf $R/gpurun_out/prof_$mode
  XPG_LOOP=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$mode -- python3 $R/bench.py --steps 400 --warmup 30 --no-cpu-baseline --no-ref-baseline --no-batched > $R/gpurun_out/bench_$mode.log 2>&1
  echo "== $mode: $(tail -1 $R/gpurun_out/bench_$mode.log | cut -c1-90)"
  f=$(find $R/gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1)
  head -5 $f | cut -c1-150 | grep xpg
done
