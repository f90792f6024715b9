cd $GRAFT_REPO_ROOT
mkdir -p tools/_build gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DXPG_STAMPS -o tools/_build/libxpoly_stamps.so xpoly_amd/csrc/xpoly_amd.hip
XPG_SO_PATH=$PWD/tools/_build/libxpoly_stamps.so PYTHONPATH=$PWD python tools/lab/probe_fastloop.py ${1:-8192} 2>&1 | tee gpurun_out/fastloop.log
XPG_SO_PATH=$PWD/tools/_build/libxpoly_stamps.so PYTHONPATH=$PWD python tools/lab/probe_fastloop.py 256 2>&1 | tee -a gpurun_out/fastloop.log
