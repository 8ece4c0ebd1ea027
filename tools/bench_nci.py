"""NCI generate alone at t5-base shapes (for rocprofv3): python tools/bench_nci.py [nq] [batch] [M] [K] [table GiB]"""
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

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = int(sys.argv[4]) if len(sys.argv) > 4 else 32
model, tower, g, rn = synth.build(dev, M, K, batch)
if len(sys.argv) > 5:
    model.prefix_table_bytes = int(float(sys.argv[5]) * (1 << 30))
ids, mask = synth.query_ids(nq, dev, np.random.default_rng(0))


def gen_all():
    return [model.generate(ids[a:a + batch], mask[a:a + batch], num_beams=10)[0] for a in range(0, nq, batch)]


gen_all()
torch.cuda.synchronize()
if os.environ.get("TRACE_GAP"):      # tools/trace_tail.py cuts the kernel trace at this idle gap: the timed pass alone
    time.sleep(0.5)
t = time.perf_counter()
gen_all()
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"nci gen: {nq} queries (batch {batch}) in {dt*1e3:.1f} ms -> {nq/dt:.0f} q/s")
