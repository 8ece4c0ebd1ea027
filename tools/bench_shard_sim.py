#!/usr/bin/env python3
"""Per-rank cost of the row-sharded dense search, simulated on ONE GPU: rank 0's shard of a world-W job,
searched with the round-1 truncated list length and with full k.  (The all-gather + merge are timed on
stand-in lists of the right shape.)  Output: one JSON line per world size."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense  # noqa: E402

dev = torch.device("cuda", 0)
nq, k = bench.N_QUERIES, bench.TOPK
query = bench.gen_queries(nq, dev, bench.N_DOCS)
for world in (8, 4, 2, 1):
    a, b = dense.shard_range(bench.N_DOCS, 0, world)
    docs = bench.gen_shard(a, b, dev, bench.N_DOCS)
    index = dense.DenseIndex(docs)
    kl = dense.truncated_list_len(k, world)
    out = {"world": world, "rows": b - a, "k_local": kl}
    for name, kk in (("trunc_ms", kl), ("full_ms", k)):
        index.search(query, kk, id_offset=a)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            s, i = index.search(query, kk, id_offset=a)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t) / 3 * 1e3
    s, i = index.search(query, kl, id_offset=a)
    all_s = s.unsqueeze(0).repeat(world, 1, 1).contiguous()
    all_i = i.unsqueeze(0).repeat(world, 1, 1).contiguous()
    dense.merge_truncated(all_s, all_i, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        dense.merge_truncated(all_s, all_i, k)
    torch.cuda.synchronize()
    out["merge_ms"] = (time.perf_counter() - t) / 3 * 1e3
    print(json.dumps(out), flush=True)
    del index, docs
    torch.cuda.empty_cache()
