# A/B of the 32-stage pass instance (XPG_BLK_ROWS: 162 = <16,2,32>, 322 = <32,2,32>, 164 = <16,4,32>) on the large shapes of the
# shapes leg -- on the hooks build (the knob is hook-only)
cd $GRAFT_REPO_ROOT
export XPG_SO_PATH=$GRAFT_REPO_ROOT/xpoly_amd/libxpoly_amd_hooks.so
for s in "8192 8192" "16384 2048" "1024 20480" "4096 8192"; do
  for r in 162 322 164; do
    echo "== $s rows_env $r"; XPG_BLK_ROWS=$r timeout 300 python tools/lab/probe_shapes.py $s 1024 2>&1 | tail -1
  done
done
