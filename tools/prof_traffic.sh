#!/bin/bash
# HBM-side traffic of the filter kernel for the default bench workload (PMC, own passes, kernel trace only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-seq2seq-legs > $OUT/tcc.log 2>&1
echo "tcc pass rc=$?"
tail -n 1 $OUT/fetch.log | cut -c1-200
