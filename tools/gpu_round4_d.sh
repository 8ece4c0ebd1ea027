#!/bin/bash
# kernel trace of the dense bench for both instruction shapes (per-launch durations of the filter)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for shape in 32 16; do
  OUT=$R/gpurun_out/r4d/trace_$shape
  rm -rf $OUT; mkdir -p $OUT
  export MEVI_IP_FILTER_MFMA=$shape
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/bench.log 2>&1
  python3 $R/tools/filter_launches.py $OUT
  f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-150
  cp $f $R/gpurun_out/r4d/kernel_stats_$shape.csv
done
