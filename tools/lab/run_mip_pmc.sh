# round 5: issue counters of the device tree walk (k_mip_tree<R32>) on the MIP leg's batch of 1024 knapsacks; one rocprofv3 pass
# per counter, the program itself behind `--` -> gpurun_out/r5mip/pmc_mip_issue.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5mip
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR VALUBusy SALUBusy SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_mip_batch.py > $O/$c.log 2>&1 || echo "$c: not collected" >> $O/skipped.txt
done
cd $R && python3 - <<'PY'
import csv, glob, json, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r5mip")
res = {"command": "rocprofv3 --pmc <counter> -- python3 tools/lab/probe_mip_batch.py (1024 0-1 knapsacks of 24 variables, 5 launches; one pass per counter: tools/lab/run_mip_pmc.sh)", "kernels": {}}
for d in sorted(glob.glob(os.path.join(O, "pmc_*"))):
    c = os.path.basename(d)[4:]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            k = row["Kernel_Name"].split("(")[0]
            if "mip" not in k and "batch" not in k:
                continue
            res["kernels"].setdefault(k, {}).setdefault(c, []).append(float(row["Counter_Value"]))
log = open(os.path.join(O, "SQ_INSTS_VALU.log")).read() if os.path.exists(os.path.join(O, "SQ_INSTS_VALU.log")) else ""
res["probe"] = [dict(ms=float(m.group(1)), nodes=int(m.group(2))) for m in re.finditer(r"ms ([\d.]+) nodes (\d+)", log)]
if os.path.exists(os.path.join(O, "skipped.txt")):
    res["not_collected"] = open(os.path.join(O, "skipped.txt")).read().split("\n")
summ = {}
for k, cs in res["kernels"].items():
    summ[k] = {c: round(sum(v) / len(v), 2) for c, v in cs.items()}
    summ[k]["launches"] = max(len(v) for v in cs.values())
res["mean_per_launch"] = summ
json.dump(res, open(os.path.join(O, "pmc_mip_issue.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)); print(res["probe"], res.get("not_collected"))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete 2>/dev/null
