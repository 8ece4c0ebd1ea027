#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench (run on the GPU box via gpurun).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stats
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/bench.log 2>&1
tail -n 1 $OUT/bench.log
find $OUT -name "*stats*.csv" | head
