#!/usr/bin/env python3
"""Randomised cross-check of the indexed dense search (f16 pre-filter, sampled last launch, exclusion, measured bound, second pass,
fallbacks) against the exact-f32 path (itself held to the oracle by tests/test_dense_gpu.py) over random shapes and row orders:
  python tools/stress_dense.py [seconds] [seed]
Every case must agree bit for bit (ids and score bits); prints one line per case and a summary of which mechanisms fired."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402
from mevi_amd import dense, hip  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
L = hip.lib()
t0 = time.time()
n_cases, fired = 0, {"second_pass": 0, "fallback": 0, "few_launches": 0, "i8": 0, "i8_open": 0}
while time.time() - t0 < budget:
    dim = int(rng.choice([64, 128, 192, 256, 768]))
    nd = int(rng.choice([3000, 40_000, 300_000, 1_200_000, 2_500_000]) * rng.uniform(0.6, 1.4))
    if dim == 768:
        nd = min(nd, 1_500_000)
    nq = int(rng.choice([1, 7, 32, 33, 64, 100, 128, 200, 300, 1500]))
    if os.environ.get("STRESS_BIG"):          # large batches with long lists: the chunk schedule's cost model picks growth 1 (12+ launches)
        nq, dim = int(rng.choice([3000, 5000, 7000])), 64
        nd = int(rng.choice([300_000, 1_000_000]) * rng.uniform(0.6, 1.4))
    if nq > 300:
        nd = min(nd, 400_000) if not os.environ.get("STRESS_BIG") else nd
    k = int(rng.choice([1, 10, 100, 257, 1000])) if not os.environ.get("STRESS_BIG") else int(rng.choice([257, 1000]))
    kind = str(rng.choice(["iid", "clustered", "duplicates", "ance_scale", "sorted_up", "sorted_down", "few_distinct",
                           "col_row_scaled", "sparse", "integers"]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    if kind in synth.CORPUS_KINDS:
        docs, info = synth.corpus(kind, dev, nd, dim, block=1 << 16, n_clusters=int(rng.choice([20, 300])))
        q, _ = synth.corpus_queries(kind, docs, nq, info, seed=int(rng.integers(1 << 30)))
    else:
        docs = torch.randn((nd, dim), device=dev, generator=g)
        q = torch.randn((nq, dim), device=dev, generator=g)
        if kind == "few_distinct":
            docs = docs[:50][torch.randint(0, 50, (nd,), device=dev, generator=g)].contiguous()
        elif kind == "col_row_scaled":      # log-normal column and row scales, a common component, a few extreme rows (the 8-bit image's scales)
            docs *= torch.exp(2.0 * torch.randn((1, dim), device=dev, generator=g))
            docs *= torch.exp(1.0 * torch.randn((nd, 1), device=dev, generator=g))
            docs += 3.0 * torch.randn((1, dim), device=dev, generator=g)
            docs[torch.randint(0, nd, (5,), device=dev, generator=g)] *= 1e3
            q *= torch.exp(2.0 * torch.randn((nq, 1), device=dev, generator=g))
        elif kind == "sparse":              # 95 % zeros: rows with a handful of coordinates (coarse steps, small rho)
            docs *= (torch.rand((nd, dim), device=dev, generator=g) < 0.05)
            q *= (torch.rand((nq, dim), device=dev, generator=g) < 0.3)
        elif kind == "integers":            # small integers: products and sums exact in every precision, ties everywhere
            docs = torch.randint(-3, 4, (nd, dim), device=dev, generator=g).float()
            q = torch.randint(-2, 3, (nq, dim), device=dev, generator=g).float()
        else:
            sc = docs @ q[0]
            docs = docs[torch.argsort(sc, descending=(kind == "sorted_down"))].contiguous()
    es, ei = dense.ip_topk(q, docs, k)
    index = dense.DenseIndex(docs)
    if os.environ.get("MEVI_IP_I8_MIN_ROWS") == "0":    # drive the <= 32-query cases through the 8-bit image (built on demand otherwise)
        index.prepare_small()
    s, i = index.search(q, k)
    torch.cuda.synchronize()
    st = hip.IpTopkStats()
    L.mevi_ip_topk_get_stats(st)
    ok = bool(torch.equal(i, ei) and torch.equal(s.view(torch.int32), es.view(torch.int32)))
    fired["second_pass"] += int(st.n_second_pass_queries > 0)
    fired["fallback"] += int(st.n_failed_queries > 0)
    fired["i8"] += int(st.n_i8_queries > 0)
    fired["i8_open"] += int(st.n_i8_unproven > 0)
    n_cases += 1
    print(f"{'ok ' if ok else 'BAD'} nq {nq:5d} nd {nd:8d} dim {dim:4d} k {k:5d} {kind:12s} launches {int(st.n_chunks):2d} second-pass {int(st.n_second_pass_queries):4d} "
          f"fallback {int(st.n_failed_queries):4d} err/bound {st.max_err_ratio:.3f}" + (f" i8 {int(st.n_i8_queries)} open {int(st.n_i8_unproven)}" if st.n_i8_queries else ""), flush=True)
    if not ok:
        bad = (i != ei).any(1).nonzero().view(-1)[:5].tolist()
        print("   first differing queries:", bad)
        sys.exit(1)
    del docs, q, s, i, es, ei, index
    torch.cuda.empty_cache()
print(f"{n_cases} cases, all identical to the exact-f32 path; searches with a second pass {fired['second_pass']}, with an exact fallback {fired['fallback']}; through the 8-bit image {fired['i8']}, of which repeated through the f16 image {fired['i8_open']}")
