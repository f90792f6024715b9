cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
oracle/_ref/dropin_demo > gpurun_out/r2g/dropin.log 2>&1; echo "dropin rc=$?"; tail -5 gpurun_out/r2g/dropin.log
timeout 900 python -m pytest tests/test_gpu_lineq.py tests/test_gpu_parity.py tests/test_gpu_mip.py -x -q > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2g/pytest.log
