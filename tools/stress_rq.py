#!/usr/bin/env python3
"""Randomised cross-check of the matrix-core RQ encoder (f16 MFMA shortlist + exact re-check, csrc/rq_fast.hip) against the exact
f32 kernel (csrc/rq_encode.hip, itself held to the oracle by tests/test_rq_gpu.py) over random shapes, corpora and codebooks:
  python tools/stress_rq.py [seconds] [seed]
Codes must be identical in every case."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from mevi_amd import rq  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
rq.KEEP_ENCODE_WORKSPACE = True
t0, n_cases, paths, amb = time.time(), 0, {}, 0
while time.time() - t0 < budget:
    dim = int(rng.choice([32, 64, 96, 128, 256, 512, 768, 1024]))
    n = int(rng.choice([1, 300, 5000, 70_000, 400_000]) * rng.uniform(0.5, 1.5)) or 1
    M = int(rng.integers(1, 9))
    K = int(rng.choice([3, 4, 16, 32, 40, 64, 100, 128, 256]))
    kind = str(rng.choice(["iid", "clustered", "ance_scale", "duplicates", "integers"]))
    book_kind = str(rng.choice(["random", "trained", "rows", "ties"]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    if os.environ.get("STRESS_VERBOSE"):
        print(f"... n {n} dim {dim} M {M} K {K} {kind} {book_kind}", flush=True)
    if kind == "integers":
        x = torch.randint(-4, 5, (n, dim), device=dev, generator=g).float()
    else:
        x, _ = synth.corpus(kind, dev, n, dim, block=1 << 16, n_clusters=int(rng.choice([10, 200])))
    if book_kind == "trained" and n >= 4 * K:
        cb, _ = rq.train_rq_codebook(x[: min(n, 20000)].contiguous(), M, K, seed=int(rng.integers(1 << 20)), n_init=1, max_iter=5)
    elif book_kind == "rows" and n >= K:
        cb = torch.stack([x[torch.randint(0, n, (K,), device=dev, generator=g)] / (1 + j) for j in range(M)])     # centroids ON data rows
    else:
        scale = float(x.std()) if n > 1 else 1.0
        cb = torch.stack([torch.randn((K, dim), device=dev, generator=g) * scale / (1 + j) for j in range(M)])
        if kind == "integers":
            cb = cb.round()
        if book_kind == "ties" and K >= 4:
            cb[:, K // 2] = cb[:, 1]                                       # duplicated centroid: the lowest index must win
    cb = cb.contiguous()
    fast = rq.rq_encode(x, cb, mode="fast")
    torch.cuda.synchronize()
    st = rq.last_encode_stats()
    if os.environ.get("STRESS_VERBOSE"):
        print("    fast done", flush=True)
    exact = rq.rq_encode(x, cb, mode="exact")
    torch.cuda.synchronize()
    ok = bool(torch.equal(fast, exact))
    paths[st.get("path")] = paths.get(st.get("path"), 0) + 1
    amb += int(st.get("ambiguous_row_levels", 0) > 0)
    n_cases += 1
    print(f"{'ok ' if ok else 'BAD'} n {n:7d} dim {dim:4d} M {M} K {K:3d} {kind:10s} codebook {book_kind:7s} path {st.get('path')} records {st.get('records', 0)} "
          f"re-encoded {st.get('rows_reencoded_exactly', 0)}", flush=True)
    if not ok:
        print("   rows differing:", int((fast != exact).any(1).sum()))
        sys.exit(1)
    del x, cb, fast, exact
print(f"{n_cases} cases, codes identical; paths {paths}; cases with ambiguous row-levels {amb}")
