#!/usr/bin/env python3
"""Per-launch times of one small search through the 8-bit image (and through the f16 image): where the time beyond the stream goes.
  MEVI_IP_TOPK_TRACE=1 python tools/probe_i8_launches.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense, hip  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
docs = bench.gen_shard(0, n, dev, n)
index = dense.DenseIndex(docs).prepare_small()
query = bench.gen_queries(64, dev, n)
L = hip.lib()
for nq, k in ((1, 10), (1, 1000), (32, 100), (32, 1000)):
    for image in ("int8", "f16"):
        index._i8_open = {} if image == "int8" else {k: 1.0}
        q = query[:nq].contiguous()
        index.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            index.search(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"--- {nq} queries top-{k} through the {image} image: {ms:.3f} ms per search; launches of one search:", flush=True)
        L.mevi_ip_topk_set_profiling(1)
        index.search(q, k)
        torch.cuda.synchronize()
        st = hip.IpTopkStats()
        L.mevi_ip_topk_get_stats(st)
        L.mevi_ip_topk_set_profiling(0)
        sys.stderr.flush()
        print(f"    filter {st.filter_ms:.3f} ms + compaction {st.compact_ms:.3f} ms in {st.n_chunks} launches", flush=True)
