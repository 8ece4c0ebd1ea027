"""One whole-corpus RQ encode for a kernel trace (TRACE_GAP=1: a pause before the timed call, tools/trace_tail.py cuts there):
python tools/rq_one.py [M] [K] [rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip as _hip  # noqa: E402
if os.environ.get("MEVI_PROBE_LIB"):  # A/B timing of another build of the library on the same device
    _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
import bench  # noqa: E402
from mevi_amd import rq  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else bench.N_DOCS
dev = torch.device("cuda", 0)
docs = bench.gen_shard(0, n, dev, n)
g = torch.Generator(device=dev).manual_seed(5)
cb = torch.stack([torch.randn((K, bench.DIM), device=dev, generator=g) * (0.05 / (1 + j)) for j in range(M)])
for _ in range(3):
    rq.rq_encode(docs, cb)
torch.cuda.synchronize()
if os.environ.get("TRACE_GAP"):
    time.sleep(0.5)
ts = []
for _ in range(1 if os.environ.get("TRACE_GAP") else 5):
    t = time.perf_counter()
    codes = rq.rq_encode(docs, cb)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
rq.KEEP_ENCODE_WORKSPACE = True
rq.rq_encode(docs, cb)
print("rq encode (%d, %d) of %d rows: %s ms; stats %s" % (M, K, n, [round(x, 2) for x in ts], rq.last_encode_stats()))
