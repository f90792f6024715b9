"""Prints the per-launch kernel times of tools/lab/run_lineq_pack.sh's two rocprofv3 traces side by side."""
import csv, glob, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lineq"
out = {}
for L in (64, 0):
    f = max(glob.glob(os.path.join(d, "prof_L%d" % L, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    out[L] = [(r["Kernel_Name"].split("(")[0][5:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
               r["Workgroup_Size_X"] + "x" + r["Workgroup_Size_Y"]) for r in rows if "xpg::k_" in r["Kernel_Name"]]
for a, b in zip(out[64], out[0]):
    print("%-16s one system per wave %9.1f us   default %9.1f us (%s)  x%.2f" % (a[0], a[1], b[1], b[2], a[1] / b[1]))
