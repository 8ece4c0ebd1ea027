#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests/test_rq_gpu.py tests/test_dense_gpu.py tests/test_e2e_gpu.py tests/test_full_size_gpu.py -x -q > gpurun_out/r3c/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -8 gpurun_out/r3c/pytest1.log
timeout 900 python tools/bench_rq.py 8841823 gpurun_out/r3c/rq.json > gpurun_out/r3c/rq.log 2>&1; echo "rq rc=$?"; grep -v "^[EW]2026" gpurun_out/r3c/rq.log | tail -3
SMALL=1 REPS=3 timeout 300 python tools/shard_w8_profile.py > gpurun_out/r3c/small.txt 2>&1; grep -v "^[EW]2026" gpurun_out/r3c/small.txt | tail -11
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3c/prof -- python3 $R/tools/bench_rq.py 8841823 > $R/gpurun_out/r3c/proflog.txt 2>&1
F=$(find $R/gpurun_out/r3c/prof -name "*kernel_stats.csv" | head -1); cut -c1-150 $F | head -12
