"""Dense arm, 33 .. 255 queries: ms per search against the per-query slot count and the chunk growth of the candidate areas
(tuning hooks MEVI_IP_TOPK_MIN_SLOTS / MEVI_IP_TOPK_GROWTH_DIV, read per call), with the lists compared to the default's.
    python tools/sweep_small_slots.py [rows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import dense  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
docs = bench.gen_shard(0, n, dev, n)
index = dense.DenseIndex(docs)
query = bench.gen_queries(bench.N_QUERIES, dev, n)


def setenv(slots, div):
    for k, v in (("MEVI_IP_TOPK_MIN_SLOTS", slots), ("MEVI_IP_TOPK_GROWTH_DIV", div)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


for bs in (64, 128, 255):
    for kk in (1000, 100):
        ref = None
        for slots, div in ((None, None), (4096, 2), (8192, None), (8192, 2), (16384, None), (16384, 2), (16384, 1.5)):
            setenv(slots, div)
            q = query[:bs].contiguous()
            s0, i0 = index.search(q, kk)
            torch.cuda.synchronize()
            if ref is None:
                ref = (s0.clone(), i0.clone())
            same = bool(torch.equal(ref[0], s0) and torch.equal(ref[1], i0))
            t = time.perf_counter()
            for r in range(10):
                index.search(query[r * bs:(r + 1) * bs].contiguous(), kk)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 10 * 1e3
            print("batch %3d top-%-4d slots %-7s growth_div %-5s %.3f ms  (%.3f of HBM peak)  lists %s" % (
                bs, kk, slots or "default", div or "default", ms, n * 768 * 2 / ms / 1e6 / 8000.0, "same" if same else "DIFFERENT"), flush=True)
setenv(None, None)
