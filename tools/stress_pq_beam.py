#!/usr/bin/env python3
"""Randomised cross-check of ProductQuantization.beam_search (the doc_multiclus > 1 path, MEVI/pq.py:613-713: top-R code paths per
row, probability products of per-level softmax(-distance)) against oracle.rq.rq_beam_search.  Tolerance-based: probabilities within
1e-3 absolute (one f32 ulp of a distance already moves them, tests/test_rq_gpu.py), labels identical except swaps between paths whose
ORACLE probabilities are within 2e-3 relative:
  python tools/stress_pq_beam.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mevi_amd import rq  # noqa: E402
from oracle import rq as orq  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
t0, cases, rows, swaps = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    dim = int(rng.choice([32, 64, 96, 768]))
    M, bits = int(rng.integers(1, 5)), int(rng.choice([1, 2, 3, 5]))
    K = 1 << bits
    R = int(rng.choice([1, 2, 5, 10]))
    n = int(rng.choice([1, 17, 200]))
    scale = float(rng.choice([0.05, 0.3, 1.0]))
    cb = (rng.standard_normal((M, K, dim)) * scale / np.arange(1, M + 1)[:, None, None]).astype(np.float32)
    x = (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    if rng.random() < 0.3:
        x = (cb[0][rng.integers(0, K, n)] + 0.3 * scale * rng.standard_normal((n, dim))).astype(np.float32)   # rows near level-0 centroids
    pq = rq.ProductQuantization("rq", M, bits, "l2", dim, device=dev)
    pq.load_codebook(cb)
    lab, sc = pq.beam_search(torch.from_numpy(x), R, return_proba=True)
    lab, sc = lab.cpu().numpy(), sc.cpu().numpy()
    olab, osc = orq.rq_beam_search(x, cb, R)
    if lab.shape != olab.shape or np.abs(sc - osc).max() > 1e-3:
        print("BAD scores", dict(dim=dim, M=M, K=K, R=R, n=n), lab.shape, olab.shape, float(np.abs(sc - osc).max()) if lab.shape == olab.shape else None)
        sys.exit(1)
    for i in range(n):
        for j in range(lab.shape[1]):
            if (lab[i, j] == olab[i, j]).all():
                continue
            twins = [jj for jj in range(lab.shape[1]) if (lab[i, j] == olab[i, jj]).all()]
            if not twins or abs(osc[i, twins[0]] - osc[i, j]) > 2e-3 * max(abs(osc[i, j]), 1e-30) + 1e-7:
                print("BAD labels", dict(dim=dim, M=M, K=K, R=R, n=n, i=i, j=j), osc[i].tolist()[:12])
                sys.exit(1)
            swaps += 1
    rows += n
    cases += 1
print(f"{cases} random codebooks, {rows} rows: beam_search labels and probabilities within tolerance of the oracle ({swaps} near-tie swaps)")
