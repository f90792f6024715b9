"""Prints the last N dispatches of a rocprofv3 --kernel-trace csv: name, duration, gap to the previous one (us)."""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s %8.2f us   gap %7.2f" % (r["Kernel_Name"][:44], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
