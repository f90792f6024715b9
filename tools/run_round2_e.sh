cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2e/pytest.log
timeout 300 python bench.py --legs rational --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['rational'])"
