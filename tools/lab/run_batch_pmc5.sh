# rounds 3 to 5: instruction counts per pivot and busy percentages of k_batch<F64> (tools/lab/probe_batch.py: 8192 LPs of 32x64,
# dense family first, then dependence-test-like; 4 launches each) -> gpurun_out/r5batch/pmc_batch_issue.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5batch
rm -rf $O; mkdir -p $O
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS VALUBusy SALUBusy; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_batch.py > $O/$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, json, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5batch")
res = {"command": "rocprofv3 --pmc <counter> -- python3 tools/lab/probe_batch.py (8192 LPs of 32x64 per family, 4 launches each: dense first, then dependence-test-like); one pass per counter (tools/lab/run_batch_pmc5.sh)", "counters": {}}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "VALUBusy", "SALUBusy"):
    vals = []
    for f in glob.glob(os.path.join(O, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c and "k_batch" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    res["counters"][c] = vals
log = open(os.path.join(O, "SQ_INSTS_VALU.log")).read()
probe = []
for m in re.finditer(r"fam (\d) .*LPs/s (\d+) pivots/s ([\d.]+)M ms ([\d.]+)", log):
    probe.append(dict(family=int(m.group(1)), lps_per_s=int(m.group(2)), mpivots_per_s=float(m.group(3)), ms=float(m.group(4))))
res["probe"] = probe
for fam, name, sl in ((0, "dense", slice(0, 4)), (1, "dep_test_like", slice(4, 8))):
    p = [x for x in probe if x["family"] == fam][0]
    piv = p["mpivots_per_s"] * 1e6 * p["ms"] / 1e3
    res[name + "_per_pivot"] = {c: round(sum(res["counters"][c][sl]) / 4 / piv, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    res[name + "_busy_percent"] = {c: round(sum(res["counters"][c][sl]) / 4, 2) for c in ("VALUBusy", "SALUBusy")}
json.dump(res, open(os.path.join(O, "pmc_batch_issue.json"), "w"), indent=1)
print({k: v for k, v in res.items() if k.endswith("pivot") or k.endswith("percent")}, probe)
PY
find $O -name "*counter_collection.csv" -delete
