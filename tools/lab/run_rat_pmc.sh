# issue / stall counters of the rational loop's kernels (bench.py --legs rational: 6 solves x 16 pivots)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3rat
rm -rf $O; mkdir -p $O
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_IFETCH"; do
  n=$(echo $g | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --legs rational --no-cpu-baseline > $O/$n.log 2>&1
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
try:
    for row in csv.DictReader(open(sys.argv[1])):
        k = row["Kernel_Name"].split("(")[0].replace("void xpg::", "")
        if not any(s in k for s in ("k_update_r32", "k_pick", "k_prep")): continue
        a = acc[(k, row["Counter_Name"])]; a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, s) in sorted(acc.items()): print("%-28s %-24s launches %4d avg %16.0f" % (k, c, n, s / n))
except Exception as e:
    print("failed", e)
PY
done
find $O -name "*counter_collection.csv" -delete
