cd $GRAFT_REPO_ROOT
mkdir -p tools/_build
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -w -DXPG_TRACE -DXPG_ANY_INLINE=__noinline__ -o tools/_build/libxpoly_trace.so xpoly_amd/csrc/xpoly_amd.hip || exit 1
echo "=== LDS kernel"; XPG_SO_PATH=$PWD/tools/_build/libxpoly_trace.so python tools/lab/probe_case_f64.py 2>&1 | grep -E "lds|hbm|six" | head -40
echo "=== HBM path"; XPG_FORCE_DEVICE_LP=1 XPG_SO_PATH=$PWD/tools/_build/libxpoly_trace.so python tools/lab/probe_case_f64.py 2>&1 | grep -E "lds|hbm|six" | head -40
