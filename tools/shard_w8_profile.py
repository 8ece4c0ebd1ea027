#!/usr/bin/env python3
"""Rank 0's shard of a W-way job (W = SHARD_W, default 8) searched with the first-round list length: per-launch
breakdown on stderr (MEVI_IP_TOPK_TRACE=1) and, under rocprofv3 --kernel-trace --stats, per-kernel time.
Also: small query batches (1/2/4/8/64) against the FULL corpus index (item: dense_small_batch)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense, hip  # noqa: E402

dev = torch.device("cuda", 0)
W = int(os.environ.get("SHARD_W", "8"))
nq, k = bench.N_QUERIES, bench.TOPK
query = bench.gen_queries(nq, dev, bench.N_DOCS)
a, b = dense.shard_range(bench.N_DOCS, 0, W)
docs = bench.gen_shard(a, b, dev, bench.N_DOCS)
index = dense.DenseIndex(docs)
kl = dense.truncated_list_len(k, W)
L = hip.lib()
for _ in range(2):
    index.search(query, kl, id_offset=a)
torch.cuda.synchronize()
L.mevi_ip_topk_set_profiling(1)
t = time.perf_counter()
reps = int(os.environ.get("REPS", "5"))
for _ in range(reps):
    index.search(query, kl, id_offset=a)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / reps * 1e3
L.mevi_ip_topk_set_profiling(0)
st = hip.IpTopkStats()
L.mevi_ip_topk_get_stats(st)
print(json.dumps({"world": W, "rows": b - a, "k_local": kl, "search_ms": ms, "filter_ms_last": st.filter_ms,
                  "compact_ms_last": st.compact_ms, "launches_last": st.n_chunks}), flush=True)
if os.environ.get("SMALL", "0") == "1":
    del index, docs
    torch.cuda.empty_cache()
    docs = bench.gen_shard(0, bench.N_DOCS, dev, bench.N_DOCS)
    index = dense.DenseIndex(docs)
    for bs in (1, 2, 4, 8, 64):
        for kk in (100, 1000):
            q = query[:bs].contiguous()
            index.search(q, kk)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                index.search(q, kk)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 10 * 1e3
            print(json.dumps({"bs": bs, "k": kk, "ms": ms, "image_TBps": bench.N_DOCS * 768 * 2 / ms / 1e9}), flush=True)
