#!/usr/bin/env python3
"""Probe: can the serial tail of the dense search (re-score gather 4.5 ms + compaction, behind the last filter
launch) hide under filter work?  The filter is a persistent kernel of one 128 KiB-LDS / 246-VGPR workgroup per CU -- nothing
else fits beside it on a CU -- so the only overlap the hardware offers is between two INDEPENDENT searches whose launches
interleave: the queries split in G groups, each searched from its own host thread on its own stream.
  python tools/probe_dense_overlap.py [groups ...]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mevi_amd import dense  # noqa: E402

dev = torch.device("cuda", 0)
n_docs, nq, k = bench.N_DOCS, bench.N_QUERIES, bench.TOPK
docs = bench.gen_shard(0, n_docs, dev, n_docs)
query = bench.gen_queries(nq, dev, n_docs)
index = dense.DenseIndex(docs)
ref_s, ref_i = index.search(query, k)
torch.cuda.synchronize()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2], ts


ms, all_ = timed(lambda: index.search(query, k))
print("one search of %d queries: %.2f ms %s" % (nq, ms, [round(x, 2) for x in all_]), flush=True)

for G in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
    tiles = (nq + 255) // 256
    cuts = [min(nq, 256 * ((tiles * g + G - 1) // G)) for g in range(G + 1)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    outs = [None] * G

    def work(g):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[g]):
            outs[g] = index.search(query[cuts[g]:cuts[g + 1]], k)

    def split():
        cur = torch.cuda.current_stream()
        for s_ in streams:
            s_.wait_stream(cur)
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for s_ in streams:
            cur.wait_stream(s_)

    ms_g, all_g = timed(split)
    s = torch.cat([o[0] for o in outs])
    i = torch.cat([o[1] for o in outs])
    same = bool(torch.equal(i, ref_i) and torch.equal(s.view(torch.int32), ref_s.view(torch.int32)))
    print("%d query groups %s on %d threads/streams: %.2f ms %s  identical=%s" % (G, cuts, G, ms_g, [round(x, 2) for x in all_g], same), flush=True)
    # the same groups one after the other on one stream: what the split costs without any overlap
    ms_q, all_q = timed(lambda: [index.search(query[cuts[g]:cuts[g + 1]], k) for g in range(G)])
    print("%d query groups, sequential: %.2f ms %s" % (G, ms_q, [round(x, 2) for x in all_q]), flush=True)
