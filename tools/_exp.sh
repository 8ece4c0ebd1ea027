cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pstats; mkdir -p $R/gpurun_out/pstats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pstats -- python3 $R/tools/bench_passage.py 1024 512 > $R/gpurun_out/pstats/log.txt 2>&1
tail -1 $R/gpurun_out/pstats/log.txt
f=$(find $R/gpurun_out/pstats -name "*kernel_stats.csv" | head -1)
cut -c1-110 $f | head -12
