#!/usr/bin/env python3
"""Probe (round 6, batch-1 latency): would keeping the NEXT layer's weights hot in the Infinity Cache (a prefetch on a side
stream) shorten the batch-1 tower pass?  The tower's 0.89 GB of weight images do not fit the 256 MB cache, so every replay
streams them from HBM.  A tower of 3 + 3 layers (~0.2 GB) does fit: if its time PER LAYER in graph replay is clearly below the
12 + 12-layer tower's, cache-hot weights help and a prefetch one layer ahead could buy the difference; if not, the launches'
own lives (activation hand-over between XCDs, launch floor) bind and prefetching weights buys nothing.
  python tools/probe_latency_mall.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from mevi_amd import t5  # noqa: E402

dev = torch.device("cuda:0")
TW = synth.tower_weights(dev)
ids, mask = synth.query_ids(64, dev, np.random.default_rng(0))


def lat(fn, n=40):
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


rows = []
for nl in (12, 6, 3, 2, 1):
    tower = t5.TwinTower(TW, device=dev, num_layers=nl, num_decoder_layers=nl)
    ms = lat(lambda i: tower.encode_query({"input_ids": ids[i % 32:i % 32 + 1], "attention_mask": mask[i % 32:i % 32 + 1]}, graph=True))
    rows.append((nl, ms))
    del tower
    torch.cuda.empty_cache()
(n_a, t_a), (n_b, t_b) = rows[0], rows[1]
per_layer_big = (t_a - t_b) / (n_a - n_b)          # marginal ms per (encoder + decoder) layer pair between 12 and 6 layers: HBM-streamed
print("layers  ms per pass   ms per layer pair (whole pass / layers)")
for nl, ms in rows:
    print(f"{nl:6d}  {ms:10.3f}   {ms / nl:8.4f}")
print(f"marginal cost of a layer pair, 12 vs 6 layers (weights from HBM): {per_layer_big * 1e3:.1f} us")
(n_c, t_c), (n_d, t_d) = rows[2], rows[4]
print(f"marginal cost of a layer pair,  3 vs 1 layers (weights fit the 256 MB cache): {(t_c - t_d) / (n_c - n_d) * 1e3:.1f} us")
