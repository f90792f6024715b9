"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/<name>.json: HBM-side traffic per launch of the sweep kernel, corrected as
MI355X_MICROARCH.md's HBM/rocprofv3 section prescribes (counters in KiB; FETCH_SIZE doubled on
gfx950 for 16-B-per-lane streaming reads; WRITE_SIZE exact).

usage: python tools/pmc_summarise.py <dir_fetch> <dir_write> <out.json> [kernel substring] [rows cols] [bench legs]
"""
import csv
import glob
import json
import os
import sys

ALG = 2 * 4096 * 8192 * 8


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            acc = out.setdefault(name, [0, 0.0])
            acc[0] += 1
            acc[1] += float(row["Counter_Value"])
    return {k: dict(launches=n, avg_kb=s / n) for k, (n, s) in out.items()}


def main():
    dfetch, dwrite, outp = sys.argv[1:4]
    want = sys.argv[4] if len(sys.argv) > 4 else "k_pipe_sweep"
    alg = 2 * int(sys.argv[5]) * int(sys.argv[6]) * 8 if len(sys.argv) > 6 else ALG
    legs = sys.argv[7] if len(sys.argv) > 7 else "pivots"
    fe, wr = per_kernel(dfetch, "FETCH_SIZE"), per_kernel(dwrite, "WRITE_SIZE")
    kf = [k for k in fe if want in k][0]
    kw = [k for k in wr if want in k][0]
    # the sweep kernel is also launched once per solve with nothing to sweep (priming): drop the
    # influence of such launches by using the per-launch average of launches only when they are few
    read_b = fe[kf]["avg_kb"] * 1024 * 2
    write_b = wr[kw]["avg_kb"] * 1024
    res = dict(
        command_fetch="rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --legs %s --no-cpu-baseline" % legs,
        command_write="same with --pmc WRITE_SIZE (separate pass: both do not fit one TCC pass)",
        note="counters are KiB; FETCH_SIZE doubled per the gfx950 correction (16-B-per-lane streaming reads are tallied at half); WRITE_SIZE exact; averages include the one priming launch per solve that sweeps nothing",
        kernel=kf, fetch_size_kib_raw=fe[kf]["avg_kb"], write_size_kib=wr[kw]["avg_kb"],
        read_bytes_corrected=read_b, write_bytes=write_b, traffic_bytes_per_launch=read_b + write_b,
        algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=(read_b + write_b) / alg,
        per_kernel=dict(FETCH_SIZE=fe, WRITE_SIZE=wr))
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("kernel", "traffic_bytes_per_launch", "traffic_over_algorithmic")}))


if __name__ == "__main__":
    main()
