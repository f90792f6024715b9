# issue / stall counters of the rational loop's kernels (bench.py --legs rational: 6 solves x 16 pivots) ->
# gpurun_out/r4rat/pmc_rational_issue.json (copied to profiles/round4_pmc_rational_issue.json)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4rat
rm -rf $O; mkdir -p $O
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $g | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --legs rational --no-cpu-baseline > $O/$n.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void xpg::", "").replace("xpg::", "")
        if not any(s in k for s in ("k_pipe_fused_r32", "k_fused_generic", "k_pipe_sweep_r32", "k_pipe_prep", "k_update_r32", "k_pick", "k_prep")): continue
        a = acc[(k, row["Counter_Name"])]; a[0] += 1; a[1] += float(row["Counter_Value"])
per = collections.defaultdict(dict)
for (k, c), (n, s) in acc.items(): per[k][c] = s / n; per[k]["launches"] = n
out = dict(command="rocprofv3 --pmc <4 counters per pass> --output-format csv -- python3 bench.py --legs rational --no-cpu-baseline (tools/lab/run_rat_pmc.sh); averages per launch over the 16 pivot positions of 6 solves",
           units="SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs", kernels=per)
for k, d in per.items():
    if "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        if "SQ_ACTIVE_INST_VALU" in d: d["valu_busy_percent"] = round(100.0 * d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 1)
        if "SQ_WAVE_CYCLES" in d: d["waves_per_simd_avg"] = round(d["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc, 2)
        if "SQ_THREAD_CYCLES_VALU" in d and "SQ_INSTS_VALU" in d and d["SQ_INSTS_VALU"] > 0:
            d["active_lanes_per_valu_instruction"] = round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_INSTS_VALU"], 1)
json.dump(out, open(os.path.join(O, "pmc_rational_issue.json"), "w"), indent=1)
for k, d in per.items(): print(k, {a: d[a] for a in ("launches", "valu_busy_percent", "waves_per_simd_avg", "active_lanes_per_valu_instruction", "SQ_INSTS_VALU") if a in d})
PY
find $O -name "*counter_collection.csv" -delete
find $O -name "*.db" -delete
