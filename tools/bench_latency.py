"""Single-query latency (what the reference's --timing_infer_step hooks time: generate.py:245-281 at batch 1,
main_models.py:3558,3729-3732 per eval batch of 2): tower, NCI generate, at t5-base shapes, synthetic weights."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

dev = torch.device("cuda:0")
model, tower, _, _ = synth.build(dev, 4, 32, None)
tower.batch_size = None
ids, mask = synth.query_ids(64, dev, np.random.default_rng(0))
model.generate(ids[:2], mask[:2], num_beams=10)


def lat(fn, n=30):
    fn(0)
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        t = time.perf_counter()
        fn(i)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts = np.sort(np.array(ts)) * 1e3
    return f"median {ts[len(ts) // 2]:.2f} ms, p90 {ts[int(len(ts) * 0.9)]:.2f} ms"


for b in (1, 2, 8):
    for graph in (True, False):
        tag = "graph replay" if graph else "eager       "
        print(f"tower, batch {b}, {tag}:        ", lat(lambda i: tower.encode_query(
            {"input_ids": ids[i % 32:i % 32 + b], "attention_mask": mask[i % 32:i % 32 + b]}, graph=graph)))
        print(f"NCI generate, batch {b}, {tag}: ", lat(lambda i: model.generate(
            ids[i % 32:i % 32 + b], mask[i % 32:i % 32 + b], num_beams=10, graph=graph)))
