# round 6 evidence: the driver's command, kernel stats and PMC passes of the two large-tableau legs; summaries are
# copied into profiles/ afterwards (profiles/README.md names each)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof
rm -rf $O; mkdir -p $O
cd $R
T0=$(date +%s); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_pivots -- python3 $R/bench.py --steps 20 --warmup 5 --legs pivots --no-cpu-baseline > $O/ks_pivots.log 2>&1
cp $(find $O/ks_pivots -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pivots_leg.csv; head -7 $O/kernel_stats_pivots_leg.csv | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_cfg2b -- python3 $R/bench.py --legs cfg2b --no-cpu-baseline > $O/ks_cfg2b.log 2>&1
cp $(find $O/ks_cfg2b -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cfg2b_leg.csv; head -7 $O/kernel_stats_cfg2b_leg.csv | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_legs -- python3 $R/bench.py --legs batched,sharded,shapes,six_e2e,one_call,rational,mip,lineq --no-cpu-baseline > $O/ks_legs.log 2>&1
cp $(find $O/ks_legs -name "*kernel_stats.csv" | head -1) $O/kernel_stats_other_legs.csv; head -8 $O/kernel_stats_other_legs.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
for leg in pivots cfg2b; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${leg}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --legs $leg --no-cpu-baseline > $O/pmc_${leg}_$c.log 2>&1
    tail -1 $O/pmc_${leg}_$c.log | cut -c1-80
  done
done
cd $R
python3 tools/pmc_summarise.py $O/pmc_pivots_FETCH_SIZE $O/pmc_pivots_WRITE_SIZE $O/pmc_hbm_traffic_4096x8192.json k_blk_sweep_full 4096 8192 pivots
python3 tools/pmc_summarise.py $O/pmc_cfg2b_FETCH_SIZE $O/pmc_cfg2b_WRITE_SIZE $O/pmc_hbm_traffic_4096x12289.json k_blk_sweep_full 4096 12289 cfg2b
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*.db" -delete
du -sh $O
