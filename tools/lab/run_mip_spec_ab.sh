# A/B of the speculative ceiling children in the device tree walk (run through gpurun from the repo root)
for spec in 1 0 256 2048; do
  XPG_MIP_DEBUG=1 XPG_MIP_SPEC=$spec timeout 300 python bench.py --legs mip --no-cpu-baseline --no-ref-baseline > gpurun_out/b_mip_$spec.json 2> gpurun_out/b_mip_$spec.err
  grep "MIP tree walk" gpurun_out/b_mip_$spec.err | sort | uniq -c | head -4
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_mip_$spec.json").read().strip().splitlines()[-1])
    m=d["mip"]; print("XPG_MIP_SPEC=$spec", {k: m[k] for k in m if k in ("value","wall_ms","nodes_per_s","problems")}, json.dumps(m)[:900])
except Exception as e:
    print("spec $spec FAILED", e); print(open("gpurun_out/b_mip_$spec.err").read()[-1200:])
PY
done
