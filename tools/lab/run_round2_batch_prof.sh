cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2bprof
rm -rf $O; mkdir -p $O
cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --legs batched --no-cpu-baseline > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats_batched_leg.csv; head -4 $O/kernel_stats_batched_leg.csv | cut -c1-220
find $O -name "*kernel_trace.csv" -delete
for c in VALUBusy SALUBusy SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/lab/probe_batch.py > $O/pmc_$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, json, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r2bprof")
out = {}
for c in ("VALUBusy", "SALUBusy", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    vals = []
    for f in glob.glob(os.path.join(O, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c and "k_batch" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    out[c] = vals
log = open(os.path.join(O, "pmc_VALUBusy.log")).read()
fam = re.findall(r"fam (\d).*?LPs/s (\d+) pivots/s ([\d.]+)M ms ([\d.]+)", log)
res = dict(command="rocprofv3 --pmc <counter> -- python3 tools/lab/probe_batch.py (8192 LPs of 32x64 per family, 4 launches each: dense first, then dependence-test-like)",
           probe=[dict(family=int(a), lps_per_s=int(b), mpivots_per_s=float(c), ms=float(d)) for a, b, c, d in fam], counters=out)
# per-pivot instruction counts of the dependence-test-like family (launches 5..8)
try:
    piv = float(fam[1][2]) * 1e6 * float(fam[1][3]) * 1e-3
    res["dep_test_like_per_pivot"] = {k: round(out[k][-1] / piv, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    res["dep_test_like_busy_percent"] = dict(VALUBusy=round(out["VALUBusy"][-1], 2), SALUBusy=round(out["SALUBusy"][-1], 2))
    piv0 = float(fam[0][2]) * 1e6 * float(fam[0][3]) * 1e-3
    res["dense_per_pivot"] = {k: round(out[k][0] / piv0, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    res["dense_busy_percent"] = dict(VALUBusy=round(out["VALUBusy"][0], 2), SALUBusy=round(out["SALUBusy"][0], 2))
except Exception as e:
    res["error"] = str(e)
json.dump(res, open(os.path.join(O, "pmc_batch_issue.json"), "w"), indent=1)
print({k: res.get(k) for k in ("dep_test_like_per_pivot", "dep_test_like_busy_percent", "dense_per_pivot", "dense_busy_percent")})
PY
find $O -name "*counter_collection.csv" -size +1M -delete
python3 -c "
import json
d=json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['batched']['families'], d['batched'].get('n1_reference_points'), d['mip']['value'], d['rational']['value'], d['cfg2b']['value'])
"
