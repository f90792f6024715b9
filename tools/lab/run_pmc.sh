# HBM-side traffic of the sweep kernel: two separate --pmc passes (no trace domains with --pmc),
# summarised by tools/pmc_summarise.py. KERNEL names the sweep (default: the blocked loop's).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${KERNEL:-k_blk_sweep}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 160 --warmup 16 --no-cpu-baseline --no-ref-baseline --no-batched > $R/gpurun_out/pmc_$c.log 2>&1
  tail -1 $R/gpurun_out/pmc_$c.log | cut -c1-100
done
cd $R && python3 tools/pmc_summarise.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_hbm_traffic.json $K
