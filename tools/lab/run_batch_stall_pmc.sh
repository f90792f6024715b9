# stall breakdown of k_batch<F64> (probe_batch.py, dense family = first 4 launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3stall
rm -rf $O; mkdir -p $O
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_batch.py > $O/$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
f, c = sys.argv[1], sys.argv[2]
vals = []
try:
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") == c and "k_batch" in row["Kernel_Name"]: vals.append(float(row["Counter_Value"]))
    print("%-24s dense %16.0f   dep %16.0f" % (c, sum(vals[:4]) / 4, sum(vals[4:8]) / 4))
except Exception as e:
    print(c, "failed", e)
PY
done
find $O -name "*counter_collection.csv" -delete
