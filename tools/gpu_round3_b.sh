#!/bin/bash
# round 3, second GPU pass: new kernels (rq_fast, small-batch filter, packed merge, aggregate sort, row softmax)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests/test_rq_gpu.py tests/test_dense_gpu.py tests/test_consumers_gpu.py tests/test_ip_rank_gpu.py -x -q > gpurun_out/r3b/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -15 gpurun_out/r3b/pytest1.log
timeout 900 python tools/bench_rq.py 8841823 gpurun_out/r3b/rq.json > gpurun_out/r3b/rq.log 2>&1; echo "rq rc=$?"; grep -v "^[EW]2026" gpurun_out/r3b/rq.log | tail -5
DATA=aniso timeout 900 python tools/bench_rq.py 4000000 gpurun_out/r3b/rq_aniso.json > gpurun_out/r3b/rq_aniso.log 2>&1; echo "rq aniso rc=$?"; grep -v "^[EW]2026" gpurun_out/r3b/rq_aniso.log | tail -5
REPS=5 timeout 600 python tools/bench_shard_sim.py gpurun_out/r3b/shard_sim.json > gpurun_out/r3b/shard_sim.log 2>&1; echo "sim rc=$?"; grep -v "^[EW]2026" gpurun_out/r3b/shard_sim.log | tail -6
SMALL=1 REPS=3 timeout 300 python tools/shard_w8_profile.py > gpurun_out/r3b/small.txt 2>&1; grep -v "^[EW]2026" gpurun_out/r3b/small.txt | tail -11
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_rq_gpu.py --deselect tests/test_dense_gpu.py --deselect tests/test_consumers_gpu.py --deselect tests/test_ip_rank_gpu.py > gpurun_out/r3b/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -15 gpurun_out/r3b/pytest2.log
