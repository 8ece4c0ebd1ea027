#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3d
timeout 600 python -m pytest tests/test_rq_gpu.py -x -q > gpurun_out/r3d/pytest1.log 2>&1; echo "pytest1 rc=$?"; tail -4 gpurun_out/r3d/pytest1.log
timeout 900 python tools/bench_rq.py 8841823 gpurun_out/r3d/rq.json > gpurun_out/r3d/rq.log 2>&1; echo "rq rc=$?"; grep -v "^[EW]2026" gpurun_out/r3d/rq.log | tail -3
for SL in 16384 8192 4096 2048; do
echo "== small slots $SL"
MEVI_IP_TOPK_SMALL_SLOTS=$SL MEVI_IP_TOPK_TRACE=1 python3 - 2>&1 <<'PY' | grep -v "^[EW]2026" | tail -24
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mevi_amd import dense, hip
dev = torch.device("cuda", 0)
docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
index = dense.DenseIndex(docs)
q = bench.gen_queries(8, dev, bench.N_DOCS)
for k in (100, 1000):
    for _ in range(2):
        index.search(q[:1].contiguous(), k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        index.search(q[:1].contiguous(), k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 100
    hip.lib().mevi_ip_topk_set_profiling(1)
    index.search(q[:1].contiguous(), k)
    torch.cuda.synchronize()
    hip.lib().mevi_ip_topk_set_profiling(0)
    print("k", k, "ms", ms, flush=True)
PY
done
