# rocprofv3 kernel-trace of the blocked loop (XPG_BLOCK=B): per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${BS:-16}; do
  rm -rf $R/gpurun_out/prof_block$B
  XPG_LOOP=block XPG_BLOCK=$B rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_block$B -- python3 $R/bench.py --steps 2000 --warmup 50 --no-cpu-baseline --no-ref-baseline --no-batched > $R/gpurun_out/bench_block$B.log 2>&1
  echo "== B=$B: $(tail -1 $R/gpurun_out/bench_block$B.log | cut -c1-90)"
  f=$(find $R/gpurun_out/prof_block$B -name "*kernel_stats.csv" | head -1)
  head -7 $f | cut -c1-170 | grep xpg
done
