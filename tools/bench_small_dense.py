"""bench.py's dense_small_batch leg alone (faiss_search.profile's regime: ONE search of 1 .. 255 queries over the resident C2
corpus): python tools/bench_small_dense.py      MEVI_IP_FILTER_QT=256 pins the wide query tile (A/B)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
docs = bench.gen_shard(0, n, dev, n)
index = dense.DenseIndex(docs)
query = bench.gen_queries(bench.N_QUERIES, dev, n)
out = bench.dense_small_batch(dev, index, query, n)
for row in out["per_batch"]:
    print(json.dumps(row))
