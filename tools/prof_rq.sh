#!/bin/bash
# rocprofv3 kernel stats of the RQ encode bench + per-launch trace of a single-query dense search (gpurun -- bash tools/prof_rq.sh)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rq_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/bench_rq.py 8841823 > $OUT/log.txt 2>&1
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cut -c1-200 $F | head -24
cd $R
MEVI_IP_TOPK_TRACE=1 python3 - > $OUT/small_trace.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from mevi_amd import dense
dev = torch.device("cuda", 0)
docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
index = dense.DenseIndex(docs)
q = bench.gen_queries(8, dev, bench.N_DOCS)
from mevi_amd import hip
for k in (100, 1000):
    for _ in range(2):
        index.search(q[:1].contiguous(), k)
    torch.cuda.synchronize()
    hip.lib().mevi_ip_topk_set_profiling(1)
    t = time.perf_counter()
    index.search(q[:1].contiguous(), k)
    torch.cuda.synchronize()
    print("k", k, "ms", (time.perf_counter() - t) * 1e3, flush=True)
    hip.lib().mevi_ip_topk_set_profiling(0)
PY
grep -v "^[EW]2026" $OUT/small_trace.txt | tail -40
