# round 2 evidence: kernel stats + PMC passes, copied into profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2prof
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_pivots -- python3 $R/bench.py --steps 20 --warmup 5 --legs pivots --no-cpu-baseline > $O/ks_pivots.log 2>&1
cp $(find $O/ks_pivots -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pivots.csv; head -6 $O/kernel_stats_pivots.csv | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_legs -- python3 $R/bench.py --legs batched,cfg2b,rational,mip,lineq --no-cpu-baseline > $O/ks_legs.log 2>&1
cp $(find $O/ks_legs -name "*kernel_stats.csv" | head -1) $O/kernel_stats_other_legs.csv; head -8 $O/kernel_stats_other_legs.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --legs pivots --no-cpu-baseline > $O/pmc_$c.log 2>&1
  tail -1 $O/pmc_$c.log | cut -c1-80
done
cd $R && python3 tools/pmc_summarise.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_hbm_traffic.json k_blk_sweep_full
cd /tmp
for c in VALUBusy SALUBusy; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --legs rational --no-cpu-baseline > $O/pmc_$c.log 2>&1
  tail -2 $O/pmc_$c.log | cut -c1-200
done
cd $R && python3 - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r2prof")
out = {}
for c in ("VALUBusy", "SALUBusy"):
    acc = {}
    for f in glob.glob(os.path.join(O, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    out[c] = {k: dict(launches=n, avg=s / n) for k, (n, s) in acc.items()}
json.dump(out, open(os.path.join(O, "pmc_rational_busy.json"), "w"), indent=1)
print({c: {k: round(v["avg"], 2) for k, v in d.items() if "update" in k or "pick" in k or "prep" in k} for c, d in out.items()})
PY
# the row-elimination kernels: per-launch times of the probe and their VALU / SALU occupancy
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_lineq -- python3 $R/tools/lab/probe_lineq.py > $O/ks_lineq.log 2>&1
cp $(find $O/ks_lineq -name "*kernel_stats.csv" | head -1) $O/kernel_stats_lineq_probe.csv; head -5 $O/kernel_stats_lineq_probe.csv | cut -c1-150
for c in VALUBusy SALUBusy; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmcl_$c -- python3 $R/tools/lab/probe_lineq.py > $O/pmcl_$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r2prof")
out = {}
for c in ("VALUBusy", "SALUBusy"):
    acc = {}
    for f in glob.glob(os.path.join(O, "pmcl_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    out[c] = {k: dict(launches=n, avg=s / n) for k, (n, s) in acc.items()}
json.dump(out, open(os.path.join(O, "pmc_lineq_busy.json"), "w"), indent=1)
print({c: {k: round(v["avg"], 2) for k, v in d.items() if "k_" in k} for c, d in out.items()})
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
