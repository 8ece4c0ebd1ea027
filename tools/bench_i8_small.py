#!/usr/bin/env python3
"""The 8-bit image of the small-batch dense search against the f16 image: same lists (bit for bit), time per search.
  python tools/bench_i8_small.py [rows] [kinds]      (default: bench.N_DOCS rows, all four corpus kinds of tools/synth.py)
One line per (kind, queries, k): ms through the f16 image, ms through the 8-bit image, whether the 8-bit pass proved every list,
largest observed error / bound, lists identical."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
import synth  # noqa: E402
from mevi_amd import dense, hip  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else list(synth.CORPUS_KINDS)
dim = int(os.environ.get("I8_DIM", "768"))
L = hip.lib()


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


rows = []
for kind in kinds:
    docs, info = synth.corpus(kind, dev, n, dim)
    queries, _ = synth.corpus_queries(kind, docs, 64, info, seed=5)
    index = dense.DenseIndex(docs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index.prepare_small()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    for nq in (1, 8, 32):
        for k in (10, 100, 1000):
            q = queries[:nq].contiguous()
            index._i8_open = {k: 1.0}                 # f16 image
            ms16, (s16, i16) = timed(lambda: index.search(q, k), 10)
            index._i8_open = {}                      # 8-bit image
            index.search(q, k)
            st = hip.IpTopkStats()
            L.mevi_ip_topk_get_stats(st)                # (of the first search: a pass that cannot prove gives up after four)
            index._i8_open = {}
            ms8, (s8, i8) = timed(lambda: index.search(q, k), 10)
            same = bool(torch.equal(i8, i16) and torch.equal(s8.view(torch.int32), s16.view(torch.int32)))
            row = {"kind": kind, "queries": nq, "k": k, "ms_f16": round(ms16, 3), "ms_i8": round(ms8, 3), "i8_queries": int(st.n_i8_queries),
                   "i8_unproven": int(st.n_i8_unproven), "launches": int(st.n_chunks), "err_over_bound": round(st.max_err_ratio, 4), "identical": same}
            rows.append(row)
            print(json.dumps(row), flush=True)
            if not same:
                sys.exit(1)
    print(json.dumps({"kind": kind, "index8_build_ms": round(build_ms, 1), "index8_bytes": int(index.index8.numel())}), flush=True)
    del docs, index, queries
    torch.cuda.empty_cache()
if len(sys.argv) > 3:
    json.dump(rows, open(sys.argv[3], "w"), indent=1)
