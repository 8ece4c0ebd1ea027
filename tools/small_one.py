"""One small-batch search over the resident C2 corpus for a kernel timeline (TRACE_GAP=1 + TIMELINE=1 with tools/trace_tail.py):
python tools/small_one.py [batch] [k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
index = dense.DenseIndex(docs)
query = bench.gen_queries(1024, dev, bench.N_DOCS)
for r in range(5):
    index.search(query[r * bs:(r + 1) * bs].contiguous(), k)
torch.cuda.synchronize()
if os.environ.get("TRACE_GAP"):
    time.sleep(0.5)
    for r in range(3):      # the clocks come back up
        index.search(query[r * bs:(r + 1) * bs].contiguous(), k)
    torch.cuda.synchronize()
    time.sleep(0.21)
t = time.perf_counter()
index.search(query[5 * bs:6 * bs].contiguous(), k)
torch.cuda.synchronize()
print("batch %d top-%d: %.3f ms" % (bs, k, (time.perf_counter() - t) * 1e3))
