cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2full
( time python -m pytest tests -x -q -m gpu > gpurun_out/r2full/pytest.log 2>&1 ) 2>&1 | grep real; tail -3 gpurun_out/r2full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
