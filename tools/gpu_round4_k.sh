#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_latency.py 2>&1 | grep -v "amdgpu.ids" | head -4
for what in tower nci; do
  OUT=$R/gpurun_out/r4k/$what; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/prof_latency.py $what > $OUT/log.txt 2>&1
  python3 $R/tools/show_stats.py $OUT 14
done
