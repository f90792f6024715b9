cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_golden.py tests/test_gpu_mip.py tests/test_gpu_lineq.py tests/test_gpu_multi.py -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2h/pytest.log
PYTHONPATH=$PWD python tools/probe_batch.py 2>&1 | grep fam
