#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3g
timeout 600 python -m pytest tests/test_rq_gpu.py -x -q > gpurun_out/r3g/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -4 gpurun_out/r3g/pytest1.log
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3g/prof -- python3 $R/tools/bench_rq.py 8841823 $R/gpurun_out/r3g/rq.json > $R/gpurun_out/r3g/proflog.txt 2>&1
grep -v "^[EW]2026" $R/gpurun_out/r3g/proflog.txt | tail -3 | cut -c1-420
cd $R
DATA=aniso CODEBOOK=trained timeout 600 python tools/bench_rq.py 4000000 gpurun_out/r3g/rq_aniso_trained.json > gpurun_out/r3g/rq_aniso.log 2>&1; grep -v "^[EW]2026" gpurun_out/r3g/rq_aniso.log | tail -2 | cut -c1-420
timeout 900 python tools/bench_chain.py > gpurun_out/r3g/chain.log 2>&1; echo "chain rc=$?"; grep -v "^[EW]2026" gpurun_out/r3g/chain.log | tail -12 | cut -c1-600
