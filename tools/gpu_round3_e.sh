#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_rq_gpu.py tests/test_dense_gpu.py -x -q > gpurun_out/r3e/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -4 gpurun_out/r3e/pytest1.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e/prof -- python3 $R/tools/bench_rq.py 8841823 $R/gpurun_out/r3e/rq.json > $R/gpurun_out/r3e/proflog.txt 2>&1
grep -v "^[EW]2026" $R/gpurun_out/r3e/proflog.txt | tail -3
F=$(find $R/gpurun_out/r3e/prof -name "*kernel_stats.csv" | head -1); cut -c1-150 $F | head -10
