cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/one_dep
rm -rf $O; mkdir -p $O
python3 $R/tools/lab/probe_one_dep.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/lab/probe_one_dep.py > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-170
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
