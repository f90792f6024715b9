cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2batch
rm -rf $O; mkdir -p $O
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_batch.py > $O/$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
f, c = sys.argv[1], sys.argv[2]
acc = {}
try:
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != c: continue
        k = row["Kernel_Name"].split("(")[0][-40:]
        a = acc.setdefault(k, []); a.append(float(row["Counter_Value"]))
    for k, v in acc.items(): print(c, k, [round(x) for x in v])
except Exception as e:
    print(c, "failed:", e)
PY
done
tail -3 $O/SQ_INSTS_VALU.log
