#!/usr/bin/env python3
"""Probe (round 6): HIP-graph replay of the tower / beam search above 8 queries.  The eager pass of 16 .. 256 queries is
host-launch bound (~250-550 launches at ~10 us of Python + ctypes each); a graph needs fixed shapes, i.e. the PADDED layout
(32 tokens per query instead of the real ~11), so it trades host time for 3x the encoder rows.  ms per call, median of 20."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from mevi_amd import nci, t5  # noqa: E402

t5.GRAPH_MAX_ROWS = nci.GRAPH_MAX_ROWS = 512
dev = torch.device("cuda:0")
model, tower, _, _ = synth.build(dev, 4, 32, None)
tower.batch_size = None
ids, mask = synth.query_ids(1024, dev, np.random.default_rng(0))
model.generate(ids[:256], mask[:256], num_beams=10)


def med(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        t = time.perf_counter()
        fn(i)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


print("queries  tower eager  tower graph   nci eager   nci graph   (ms per call)")
for b in (8, 16, 32, 64, 128, 256):
    sl = lambda i: slice((i * b) % 512, (i * b) % 512 + b)      # noqa: E731
    row = [b]
    for what in ("tower", "nci"):
        for graph in (False, True):
            if what == "tower":
                row.append(med(lambda i: tower.encode_query({"input_ids": ids[sl(i)], "attention_mask": mask[sl(i)]}, graph=graph)))
            else:
                row.append(med(lambda i: model.generate(ids[sl(i)], mask[sl(i)], num_beams=10, graph=graph)))
    print("%7d  %11.3f  %11.3f  %10.3f  %10.3f" % tuple(row), flush=True)
# same bits?
b = 64
e = tower.encode_query({"input_ids": ids[:b], "attention_mask": mask[:b]}, graph=False)
g = tower.encode_query({"input_ids": ids[:b], "attention_mask": mask[:b]}, graph=True)
g = tower.encode_query({"input_ids": ids[:b], "attention_mask": mask[:b]}, graph=True)
print("tower graph == eager bits at 64 queries:", bool(torch.equal(e, g)), float((e - g).abs().max()))
de = model.generate(ids[:b], mask[:b], num_beams=10, graph=False)
dg = model.generate(ids[:b], mask[:b], num_beams=10, graph=True)
dg = model.generate(ids[:b], mask[:b], num_beams=10, graph=True)
print("nci graph == eager beams at 64 queries:", bool(torch.equal(de[0], dg[0])), float(np.abs(np.asarray(de[1]) - np.asarray(dg[1])).max()))
