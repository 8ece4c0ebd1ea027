cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/cstats; mkdir -p $R/gpurun_out/cstats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cstats -- python3 $R/tools/probe_dense.py 2000000 > $R/gpurun_out/cstats/log.txt 2>&1
f=$(find $R/gpurun_out/cstats -name "*kernel_stats.csv" | head -1)
cut -c1-150 $f | head -12
