#!/bin/bash
# kernel stats of the passage tower and the query tower (round 4, after the 16x16x32 GEMM)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for what in passage tower; do
  OUT=$R/gpurun_out/r4h/$what; rm -rf $OUT; mkdir -p $OUT
  if [ $what = passage ]; then args="tools/bench_passage.py 4096 2048"; else args="tools/bench_tower.py 6980"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$args > $OUT/log.txt 2>&1
  tail -n 2 $OUT/log.txt
  python3 $R/tools/show_stats.py $OUT 16
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4h/${what}_kernel_stats.csv
done
