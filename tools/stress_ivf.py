#!/usr/bin/env python3
"""Randomised cross-check of the IVF-Flat index (mevi_amd/ivf.py: `--param IVF<n>,Flat`, MEVI/faiss_search.py:13-21,89) against
oracle.dense.ivf_flat_search given the index's own centroids: list membership and results bit for bit, over random corpora (clustered,
unclustered, duplicated rows), list counts (incl. more lists than distinct rows -> empty lists), nprobe 1..nlist and k up to 1000:
  python tools/stress_ivf.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mevi_amd import ivf  # noqa: E402
from oracle import dense as odense  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    dim = int(rng.choice([32, 64, 100, 768]))
    nd = int(rng.choice([40, 800, 6000, 30000]))
    nq = int(rng.choice([1, 9, 90]))
    nlist = int(rng.choice([1, 2, 7, 16, 40, 100]))
    nprobe = int(rng.integers(1, nlist + 1))
    k = int(rng.choice([1, 30, 100, 1000]))
    kind = str(rng.choice(["clustered", "plain", "duplicates"]))
    centres = rng.standard_normal((12, dim)).astype(np.float32) * 2
    if kind == "clustered":
        d = (centres[rng.integers(0, 12, size=nd)] + rng.standard_normal((nd, dim))).astype(np.float32)
    elif kind == "plain":
        d = rng.standard_normal((nd, dim)).astype(np.float32)
    else:
        d = rng.standard_normal((max(1, nd // 40), dim)).astype(np.float32)[rng.integers(0, max(1, nd // 40), nd)]
    q = (centres[rng.integers(0, 12, size=nq)] + rng.standard_normal((nq, dim))).astype(np.float32)
    dt, qt = torch.from_numpy(d).to(dev), torch.from_numpy(q).to(dev)
    index = ivf.IVFFlatIndex(dt, nlist)
    s, i = index.search(qt, k, nprobe)
    es, ei, list_of = odense.ivf_flat_search(q, d, index.centroids.cpu().numpy(), k, nprobe)
    ok = np.array_equal(index.list_of.cpu().numpy(), list_of) and np.array_equal(i.cpu().numpy(), ei) and \
        np.array_equal(s.cpu().numpy().view(np.uint32), es.view(np.uint32))
    cases += 1
    if not ok:
        print("BAD", dict(dim=dim, nd=nd, nq=nq, nlist=nlist, nprobe=nprobe, k=k, kind=kind))
        sys.exit(1)
print(f"{cases} random IVF-Flat indexes: list membership and results identical to the oracle's")
