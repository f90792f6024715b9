# round 5: what does the pass wait for at 32 stages? tools/_build/sweep_lab3 (the "pair" variants: the product's body at 24 and
# 32 stages) under one counter per pass -> gpurun_out/sweep_pmc/summary.txt (mean per kernel name and counter)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweep_pmc
rm -rf $O; mkdir -p $O
for c in VALUBusy SALUBusy MemUnitBusy MemUnitStalled WriteUnitStalled SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVES FetchSize WriteSize L2CacheHit; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $R/tools/_build/sweep_lab3 "NB=32" > $O/$c.log 2>&1 || echo "$c: failed" >> $O/summary.txt
done
cd $R && python3 - <<'PY' >> $O/summary.txt
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "sweep_pmc")
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-22s n=%3d mean %.6g" % (k, c, len(v), sum(v) / len(v)))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/summary.txt
