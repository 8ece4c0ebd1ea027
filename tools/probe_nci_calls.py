"""Which call sites convert f32 rows to split images / gather rows during one NCI generate pass (shapes and counts):
python tools/probe_nci_calls.py [nq] [M] [K]"""
import collections
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from mevi_amd import ops  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 6980
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
model, tower, g, rn = synth.build(dev, M, K, nq)
ids, mask = synth.query_ids(nq, dev, np.random.default_rng(0))
model.generate(ids, mask, num_beams=10)
counts = collections.Counter()


def wrap(name):
    fn = getattr(ops, name)

    def w(x, *a, **k):
        fr = [f for f in traceback.extract_stack()[:-1] if "mevi_amd" in f.filename][-3:]
        where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr))
        shape = tuple(getattr(x, "shape", ()))
        counts[(name, shape, where)] += 1
        return fn(x, *a, **k)

    setattr(ops, name, w)


for n in ("split_rows", "gather_rows", "rmsnorm", "add_layernorm", "scatter_rows"):
    wrap(n)
model.generate(ids, mask, num_beams=10)
for (name, shape, where), c in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1] * (kv[0][1][0] if kv[0][1] else 0))):
    print(f"{name:14s} {str(shape):18s} x{c:3d}  {where}")
