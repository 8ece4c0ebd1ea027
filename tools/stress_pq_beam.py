#!/usr/bin/env python3
"""Randomised cross-check of ProductQuantization.beam_search (the doc_multiclus > 1 path, MEVI/pq.py:613-713: top-R code paths per
row, probability products of per-level softmax(-distance)) against a float64 restatement of it (oracle.rq.rq_beam_search is the f32 form the golden G4 pins).  A greedy / beam decision that
hangs on a margin below the f32 noise legitimately flips between summation orders (and with it the whole path), so rows with such a
margin at any level are skipped; on all others labels must be identical and probabilities within 2e-3 relative:
  python tools/stress_pq_beam.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mevi_amd import rq  # noqa: E402

def beam_search_f64(x, cb, R):
    """pq.beam_search in float64 (MEVI/pq.py:640-700) + per row the smallest RELATIVE gap between the last kept and the first dropped
    candidate (and between neighbouring kept ones) over the levels: rows whose decisions hang on less than the f32 noise are skipped."""
    x, cb = x.astype(np.float64), cb.astype(np.float64)
    M, K, dim = cb.shape
    n = x.shape[0]
    score = np.ones((n, 1))
    resid = x[:, None, :].copy()
    labels = np.zeros((n, 1, 0), np.int32)
    gap = np.full(n, np.inf)
    for j in range(M):
        diff = resid[:, :, None, :] - cb[j][None, None]
        neg = -(diff * diff).sum(-1)
        neg -= neg.max(-1, keepdims=True)
        p = np.exp(neg)
        p = score[:, :, None] * (p / p.sum(-1, keepdims=True))
        nb = p.shape[1]
        flat = p.reshape(n, nb * K)
        if R < nb * K:
            order = np.argsort(-flat, axis=1, kind="stable")
            srt = np.take_along_axis(flat, order, 1)
            g = (srt[:, :R] - srt[:, 1:R + 1]) / np.maximum(srt[:, :R], 1e-300)
            gap = np.minimum(gap, np.where(srt[:, :R] > 1e-12, g, np.inf).min(1))
            order = order[:, :R]
        else:                     # fewer candidates than beams: all of them, in (beam, code) order -- nothing is decided (pq.py:686-700)
            order = np.broadcast_to(np.arange(nb * K)[None], (n, nb * K))
        prev, code = order // K, order % K
        score = np.take_along_axis(flat, order, 1)
        labels = np.concatenate([np.take_along_axis(labels, prev[:, :, None], 1), code[:, :, None].astype(np.int32)], -1)
        if j != M - 1:
            resid = np.take_along_axis(resid, prev[:, :, None], 1) - cb[j][code]
    return labels, score, gap


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
t0, cases, rows, skipped = time.time(), 0, 0, 0
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
    olab, osc, gap = beam_search_f64(x, cb, R)
    clear = gap > 5e-3                                   # rows whose every keep / drop decision has a margin beyond f32 noise
    if lab.shape != olab.shape:
        print("BAD shape", dict(dim=dim, M=M, K=K, R=R, n=n), lab.shape, olab.shape)
        sys.exit(1)
    bad = [i for i in np.nonzero(clear)[0] if not np.array_equal(lab[i][osc[i] > 1e-9], olab[i][osc[i] > 1e-9])
           or np.abs(sc[i] - osc[i]).max() > 2e-3 * max(float(osc[i].max()), 1e-30) + 1e-6]
    if bad:
        w = bad[0]
        print("BAD", dict(dim=dim, M=M, K=K, R=R, n=n, scale=scale), "row", int(w), "gap", float(gap[w]), lab[w].tolist(), olab[w].tolist(),
              sc[w].tolist(), osc[w].tolist())
        sys.exit(1)
    skipped += int((~clear).sum())
    rows += n
    cases += 1
print(f"{cases} random codebooks, {rows} rows: beam_search labels identical and probabilities within 2e-3 of a float64 restatement on every row whose "
      f"decisions have a margin (> 5e-3 relative); {skipped} rows skipped as near-ties")
