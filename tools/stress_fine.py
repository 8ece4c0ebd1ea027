#!/usr/bin/env python3
"""Randomised cross-check of the fine stage (mevi_amd.fine.FineStage.rerank: cluster gather + pair_dot + segment sort,
csrc/rerank.hip) against oracle.dense.fine_stage (MEVI/main_models.py:3921-4013 restated: documents of the beam clusters in beam
order, a repeated cluster listed again, an absent one skipped, fmaf-chain scores, score desc / id asc) over random shapes --
empty clusters, repeated beam clusters, one huge cluster, duplicated rows (score ties):
  python tools/stress_fine.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mevi_amd.fine import FineStage  # noqa: E402
from mevi_amd.rq import ClusterIndex  # noqa: E402
from oracle import dense as odense  # noqa: E402
from oracle import rq as orq  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
t0, cases, queries = time.time(), 0, 0
while time.time() - t0 < budget:
    N = int(rng.choice([50, 3000, 40000]))
    dim = int(rng.choice([32, 64, 100, 768]))
    M, K = int(rng.integers(1, 5)), int(rng.choice([2, 6, 32]))
    B, R = int(rng.integers(1, 40)), int(rng.choice([1, 5, 10]))
    emb = rng.standard_normal((N, dim)).astype(np.float32)
    if rng.random() < 0.3:
        emb[rng.integers(0, N, N // 3)] = emb[0]                       # duplicated rows: equal scores, order by id
    codes = rng.integers(0, K, size=(N, M)).astype(np.int32)
    if rng.random() < 0.3:
        codes[: N // 2] = codes[0]                                     # one huge cluster
    cluster, _ = orq.cluster_dict(codes)
    q = rng.standard_normal((B, dim)).astype(np.float32)
    beams = rng.integers(0, K, size=(B, R, M))
    for b in range(B):
        if b % 3 == 0:
            beams[b] = codes[rng.integers(0, N, size=R)]              # populated clusters
        if b % 4 == 1 and R > 1:
            beams[b, -1] = beams[b, 0]                                # a repeated cluster
    fs = FineStage(torch.from_numpy(emb).to(dev), ClusterIndex.from_codes(codes, K))
    out, ndoc = fs.rerank(torch.from_numpy(q).to(dev), beams)
    for b in range(B):
        docs, sc, nd = odense.fine_stage(q[b], emb, cluster, beams[b])
        got_d, got_s = np.asarray(out[b][0]), np.asarray(out[b][1])
        if int(ndoc[b]) != nd or not np.array_equal(got_d, docs) or not np.array_equal(got_s.view(np.uint32), sc.view(np.uint32)):
            print("BAD", dict(N=N, dim=dim, M=M, K=K, B=B, R=R, b=b, ndoc=(int(ndoc[b]), nd)))
            sys.exit(1)
        queries += 1
    cases += 1
print(f"{cases} random inputs, {queries} queries: fine lists identical to the oracle's (ids, score bits, candidate counts)")
