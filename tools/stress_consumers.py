#!/usr/bin/env python3
"""Randomised cross-check of the device consumers (mevi_amd/consumers.py + csrc/consumers.hip: cluster ranks, score combination,
ranking) against the literal dict loop of ensemble_marco.py (tests/test_consumers_gpu.py::_dict_loop, pinned by golden G6), over
random list shapes with everything the loop is sensitive to (-1 padding, repeated documents, equal scores, fine lists longer
than the dense one, empty lists, repeated beam clusters) and random (alpha, beta, gamma):
  python tools/stress_consumers.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_consumers_gpu as T  # noqa: E402  (the generators and the reference loop)
from mevi_amd import metrics  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, cases, lists = time.time(), 0, 0
while time.time() - t0 < budget:
    nq = int(rng.integers(1, 80))
    k = int(rng.choice([1, 3, 10, 100, 1000]))
    fine_max = int(rng.choice([0, 2, 40, 3000]))
    R = int(rng.choice([2, 5, 10]))
    repeats = int(rng.integers(1, R))
    M, K = int(rng.integers(1, 5)), int(rng.choice([2, 6, 32]))
    with_fine = bool(rng.integers(0, 2))
    n_docs = max(5000, 3 * k + fine_max + 10)
    codes, dense_p, dense_s, fine_p, fine_s, beams = T._lists(rng, nq, n_docs, k, fine_max, M, K, R, repeats=repeats)
    try:
        cranks, n_clusters = metrics.cluster_ranks(dense_p, beams, metrics.ArrayMapping(codes))
    except AssertionError:         # too few distinct code paths for R - repeats clusters per query: the reference refuses such a file too
        continue
    inp = T._inputs(codes, dense_p, dense_s, fine_p, fine_s, beams, with_fine)
    assert inp.n_clusters == n_clusters
    cr = inp.ranks()
    seg, oseg = inp.seg_d.cpu().numpy(), inp.out_seg.cpu().numpy()
    for i, q in enumerate(inp.queries):
        assert cr[seg[i]:seg[i + 1]].cpu().tolist() == cranks[q], ("ranks", q)
    for _ in range(3):
        a, b, g = float(rng.choice([0.0, 0.6, 1.7, rng.uniform(0, 3)])), float(rng.choice([0.0, 0.03, 0.5, rng.uniform(0, 1)])), \
            float(rng.choice([0.0, 0.02, 0.9, rng.uniform(0, 1)]))
        out_docs, out_n = inp.ensemble(cr, a, b, g)
        out_docs, out_n = out_docs.cpu().numpy(), out_n.cpu().numpy()
        for i, q in enumerate(inp.queries):
            want = T._dict_loop(dense_p[q], dense_s[q], cranks[q], fine_p[q] if with_fine else None, fine_s[q] if with_fine else None,
                                n_clusters, a, b, g)
            got = out_docs[oseg[i]:oseg[i] + out_n[i]].tolist()
            if got != want:
                print("BAD", dict(nq=nq, k=k, fine_max=fine_max, R=R, repeats=repeats, M=M, K=K, with_fine=with_fine, a=a, b=b, g=g, q=q))
                sys.exit(1)
            lists += 1
    cases += 1
print(f"{cases} random inputs, {lists} ranked lists: the device ensemble equals the reference's dict loop everywhere")
