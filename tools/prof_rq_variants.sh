#!/bin/bash
# rocprofv3 per-kernel time of the RQ encoder's main kernels: (4,32) split and un-split (MEVI_RQ_NO_SPLIT=1), (3,256); the second
# pass runs with whatever environment switch is exported in the loop below (used for the round-3 timing experiments)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/exp1
for MODE in normal blocked; do
  if [ $MODE = blocked ]; then export MEVI_RQ_NOTHING=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp1/$MODE -- python3 - > $R/gpurun_out/exp1/$MODE.log 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from mevi_amd import rq
dev = torch.device("cuda", 0)
docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
g = torch.Generator(device=dev).manual_seed(5)
for M, K in ((4, 32), (3, 256)):
    cb = torch.stack([torch.randn((K, 768), device=dev, generator=g) * (0.05 / (1 + j)) for j in range(M)])
    for mode in (("fast",) if os.environ.get("MEVI_RQ_EXPERIMENT_BLOCKED_X") else ("fast",)):
        for _ in range(3):
            rq.rq_encode(docs, cb, mode="fast")
    torch.cuda.synchronize()
os.environ["MEVI_RQ_NO_SPLIT"] = "1"
cb = torch.stack([torch.randn((32, 768), device=dev, generator=g) * (0.05 / (1 + j)) for j in range(4)])
for _ in range(3):
    rq.rq_encode(docs, cb, mode="fast")
torch.cuda.synchronize()
PY
  F=$(find $R/gpurun_out/exp1/$MODE -name "*kernel_stats.csv" | head -1); echo "== $MODE"; grep "rq_fast_kernel\|rf_fixup" $F | cut -c1-160
done
