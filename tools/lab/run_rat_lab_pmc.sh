# SQ counters per variant of tools/lab/rat_sweep_lab.hip on one exported pivot
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3ratlab
P=${1:-12}
rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w -I $R/xpoly_amd/csrc -o /tmp/rat_sweep_lab $R/tools/lab/rat_sweep_lab.hip || exit 1
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL"; do
  n=$(echo $g | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $g --output-format csv -d $O/pmc_$n -- /tmp/rat_sweep_lab $R/tools/lab/_data/pivot$P.bin > $O/$n.log 2>&1
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
try:
    for row in csv.DictReader(open(sys.argv[1])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("xpg::", "")
        if "k_sweep" not in k: continue
        a = acc.setdefault((k, row["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    for (k, c), (n, s) in acc.items(): print("%-52s %-22s n %3d avg %14.0f" % (k[:52], c, n, s / n))
except Exception as e:
    print("failed", e)
PY
done
find $O -name "*counter_collection.csv" -delete
