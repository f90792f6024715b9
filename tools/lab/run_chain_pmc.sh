# dynamic instruction counts of the chain launch (per launch, all 130 waves): separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chainpmc
rm -rf $O; mkdir -p $O
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/bench.py --steps 2 --warmup 1 --legs pivots --no-cpu-baseline > $O/$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "chainpmc")
for c in sorted(os.listdir(O)):
    if not os.path.isdir(os.path.join(O, c)): continue
    acc = {}
    for f in glob.glob(os.path.join(O, c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    for k, (n, s) in acc.items():
        if "chain" in k or "sweep_full" in k:
            print("%-20s %-40s launches %5d  avg per launch %14.1f" % (c, k[:40], n, s / n))
PY
find $O -name "*counter_collection.csv" -delete
