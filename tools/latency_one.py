"""One graph-replayed batch-1 pass for a kernel timeline (TRACE_GAP=1 TIMELINE=1 with tools/trace_tail.py):
python tools/latency_one.py [tower|nci]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "tower"
dev = torch.device("cuda:0")
model, tower, _, _ = synth.build(dev, 4, 32, None)
ids, mask = synth.query_ids(64, dev, np.random.default_rng(0))


def run(i):
    if what == "tower":
        return tower.encode_query({"input_ids": ids[i:i + 1], "attention_mask": mask[i:i + 1]}, graph=True)
    return model.generate(ids[i:i + 1], mask[i:i + 1], num_beams=10, graph=True)


for i in range(30):
    run(i % 32)
torch.cuda.synchronize()
if os.environ.get("TRACE_GAP"):
    time.sleep(0.3)
    for i in range(20):
        run(i)
    torch.cuda.synchronize()
    time.sleep(0.21)
t = time.perf_counter()
run(3)
torch.cuda.synchronize()
print("%s batch 1, graph replay: %.3f ms" % (what, (time.perf_counter() - t) * 1e3))
