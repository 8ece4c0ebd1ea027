#!/bin/bash
# rocprofv3 kernel stats of the W=8 shard search + per-launch trace + small-batch timings (gpurun -- bash tools/prof_shard.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/shard8
rm -rf $OUT; mkdir -p $OUT
MEVI_IP_TOPK_TRACE=1 REPS=1 python3 $R/tools/shard_w8_profile.py > $OUT/trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/shard_w8_profile.py > $OUT/log.txt 2>&1
SMALL=1 REPS=3 python3 $R/tools/shard_w8_profile.py > $OUT/small.txt 2>&1
grep -v "^[EW]2026" $OUT/trace.txt | tail -n 30
grep -v "^[EW]2026" $OUT/small.txt | tail -n 12
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -20 $F
