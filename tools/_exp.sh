cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/sstats; mkdir -p $R/gpurun_out/sstats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sstats -- python3 $R/tools/bench_stages.py 2048 512 > $R/gpurun_out/sstats/log.txt 2>&1
grep "tower\|nci\|passage\|fine\|rq enc" $R/gpurun_out/sstats/log.txt
