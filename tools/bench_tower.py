"""The T5-ANCE-shaped query tower alone (for rocprofv3): python tools/bench_tower.py [nq]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mevi_amd import hip as _hip  # noqa: E402
if os.environ.get("MEVI_PROBE_LIB"):
    _hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6980
dev = torch.device("cuda:0")
tower = synth.build_tower(dev)
ids, mask = synth.query_ids(nq, dev, np.random.default_rng(0))
q = {"input_ids": ids, "attention_mask": mask}
tower.encode_query(q)
torch.cuda.synchronize()
if os.environ.get("TRACE_GAP"):      # tools/trace_tail.py cuts the kernel trace at this idle gap: the timed pass alone
    time.sleep(0.5)
t = time.perf_counter()
for _ in range(3):
    tower.encode_query(q)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print(f"tower: {nq} queries ({int(mask.sum())} real tokens) in {dt*1e3:.1f} ms -> {nq/dt:.0f} q/s")
