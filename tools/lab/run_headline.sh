# the two blocked-loop legs of bench.py (self-checked against the reference fixture) and, if a stamped build travelled
# along, the chain's per-stage timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --legs pivots,cfg2b --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/headline.json 2> gpurun_out/headline.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/headline.json").read().strip().split("\n")[-1])
print("pivots/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "chain", d["roofline"].get("chain"), "loop", d["roofline"].get("loop_effective"))
print("cfg2b", {k: v for k, v in d.items() if "cfg2b" in k})
print("self_check", d.get("self_check"))
PY
tail -3 gpurun_out/headline.err
if [ -f tools/_build/libxpoly_stamps.so ]; then bash tools/lab/run_chain_stamps.sh | grep -E "mean|stage-to" ; fi
