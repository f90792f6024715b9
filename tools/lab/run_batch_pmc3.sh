# instruction counts per pivot of k_batch<F64> (probe_batch.py: dense family first, then dependence-test-like)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch
rm -rf $O; mkdir -p $O
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS VALUBusy SALUBusy; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_batch.py > $O/$c.log 2>&1
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
f, c = sys.argv[1], sys.argv[2]
acc = {}
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") != c: continue
    k = row["Kernel_Name"].split("(")[0][-30:]
    acc.setdefault(k, []).append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "k_batch" in k: print(c, k, [round(x, 2) for x in v])
PY
done
grep "pivots/s" $O/SQ_INSTS_VALU.log
find $O -name "*counter_collection.csv" -delete
