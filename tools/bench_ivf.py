"""IVF-Flat (faiss_search.py --param IVF<n>,Flat) at C2 size on the synthetic corpus: build / search time and recall vs exact.
python tools/bench_ivf.py [nlist] [nprobe,...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense, ivf  # noqa: E402

nlist = int(sys.argv[1]) if len(sys.argv) > 1 else 100
probes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,4,16").split(",")]
dev = torch.device("cuda:0")
docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
q = bench.gen_queries(bench.N_QUERIES, dev, bench.N_DOCS)
torch.cuda.synchronize()
t = time.perf_counter()
index = ivf.IVFFlatIndex(docs, nlist)
torch.cuda.synchronize()
print(f"IVF{nlist},Flat build (k-means {ivf.NITER} it. on {min(bench.N_DOCS, 256 * nlist)} rows + assign + list-major copy): {time.perf_counter() - t:.2f} s")
es, ei = dense.DenseIndex(docs).search(q, bench.TOPK)
for p in probes:
    index.search(q, bench.TOPK, p)
    torch.cuda.synchronize()
    t = time.perf_counter()
    s, i = index.search(q, bench.TOPK, p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rec = {}
    for a in range(0, q.shape[0], 256):
        for c, v in ivf.recall_report(i[a:a + 256], ei[a:a + 256]).items():
            rec[c] = rec.get(c, 0.0) + v * min(256, q.shape[0] - a) / q.shape[0]
    print(f"nprobe {p:3d}: {dt * 1e3:8.1f} ms  {q.shape[0] / dt:9.0f} q/s   recall vs exact " + ", ".join(f"@{c} {v:.4f}" for c, v in rec.items()))
