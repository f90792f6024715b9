cd $GRAFT_REPO_ROOT
mkdir -p tools/_build gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -w -DXPG_LIFE -DXPG_ANY_INLINE=__noinline__ -o tools/_build/libxpoly_life.so xpoly_amd/csrc/xpoly_amd.hip
XPG_SO_PATH=$PWD/tools/_build/libxpoly_life.so PYTHONPATH=$PWD python tools/lab/probe_batch_life.py ${1:-1} ${2:-768} ${3:-} 2>&1 | tee gpurun_out/batch_life.log
