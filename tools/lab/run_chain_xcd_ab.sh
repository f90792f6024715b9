# A/B of the blocked loop's forms on the bench LPs (run through gpurun from the repo root)
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --legs pivots,cfg2b --steps 10 --warmup 3 --no-cpu-baseline --no-ref-baseline > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/b_$name.json").read().strip().splitlines()[-1])
    print("$name", "pivots/s", d["value"], "us/pivot", d["roofline"]["loop_effective"]["us_per_pivot"], "sweep us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], "cfg2b", d.get("cfg2b",{}).get("value"), d.get("cfg2b",{}).get("roofline",{}).get("avg_launch_us"), str(d.get("self_check"))[-40:])
except Exception as e:
    print("$name FAILED", e); print(open("gpurun_out/b_$name.err").read()[-1500:])
PY
}
run fold XPG_NOOP=1
run nofold XPG_CHAIN_FOLD=0
run b24 XPG_BLOCK=24
run b16 XPG_BLOCK=16
python -m pytest tests/test_gpu_large_golden.py tests/test_gpu_edges.py tests/test_gpu_multi.py tests/test_gpu_sane_mode.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 2>&1 | tail -25
