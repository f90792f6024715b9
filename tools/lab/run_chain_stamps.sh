# the chain's per-stage timeline from a -DXPG_STAMPS build made in the authoring container (tools/_build/libxpoly_stamps.so)
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=${XPG_SO_PATH:-$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so}   # hook-only knobs: the -DXPG_TEST_HOOKS build
mkdir -p gpurun_out
XPG_SO_PATH=$PWD/tools/_build/libxpoly_stamps.so PYTHONPATH=$PWD python tools/lab/probe_chain_ts.py 2>&1 | tee gpurun_out/chain_stamps.log
