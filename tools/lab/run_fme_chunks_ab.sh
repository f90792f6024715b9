# A/B of the two-stream chunk pipeline of the packed fme entry point (run through gpurun from the repo root)
timeout 600 python -m pytest tests/test_gpu_lineq.py tests/test_gpu_ragged.py -m gpu -x -q 2>&1 | tail -5
for ch in 0 1 4 8 16; do
  XPG_FME_CHUNKS=$ch timeout 300 python bench.py --legs lineq --no-cpu-baseline --no-ref-baseline > gpurun_out/b_fme_$ch.json 2> gpurun_out/b_fme_$ch.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_fme_$ch.json").read().strip().splitlines()[-1])
    print("XPG_FME_CHUNKS=$ch", [(s["rows"], s["cols"], s.get("fme_host_arrays_systems_per_s"), s.get("fme_host_arrays_pcie_bound_systems_per_s")) for s in d["lineq"]["shapes"]])
except Exception as e:
    print("chunks $ch FAILED", e); print(open("gpurun_out/b_fme_$ch.err").read()[-1200:])
PY
done
