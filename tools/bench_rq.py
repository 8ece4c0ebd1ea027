#!/usr/bin/env python3
"""RQ encode of the C2 corpus (8,841,823 x 768 f32): the matrix-core encoder with exact re-check (csrc/rq_fast.hip) next to
the exact f32 VALU kernel (csrc/rq_encode.hip), at (M, K) = (4, 32) [the scripts] and (3, 256) [BASELINE configs[2]].
Codes of both must be identical (checked on the whole corpus); roofline = SURVEY 8(d)'s 4 N d + 4 N M bytes / time.
  python tools/bench_rq.py [rows] [out.json]        DATA=aniso : rows share a large common component (dense-retriever-like)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mevi_amd import rq  # noqa: E402

rq.KEEP_ENCODE_WORKSPACE = True

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_DOCS
docs = bench.gen_shard(0, n, dev, n)
if os.environ.get("DATA") == "aniso":
    g = torch.Generator(device=dev).manual_seed(9)
    docs += 0.4 * torch.randn((1, bench.DIM), device=dev, generator=g)
g = torch.Generator(device=dev).manual_seed(5)
out = {"rows": n, "dim": bench.DIM, "data": os.environ.get("DATA", "bench"), "device": torch.cuda.get_device_name(0)}
for M, K in ((4, 32), (3, 256)):
    if os.environ.get("CODEBOOK") == "trained":
        cb, _ = rq.train_rq_codebook(docs[:200_000].contiguous(), M, K, seed=1)
    else:
        cb = torch.stack([torch.randn((K, bench.DIM), device=dev, generator=g) * (0.05 / (1 + j)) for j in range(M)])
        if os.environ.get("DATA") == "aniso":
            cb[0] += docs[:4096].mean(0, keepdim=True)
    res = {}
    for mode in ("fast", "exact"):
        rq.rq_encode(docs[:1 << 16], cb, mode=mode)
        torch.cuda.synchronize()
        reps = 3 if mode == "fast" else 1
        t = time.perf_counter()
        for _ in range(reps):
            codes = rq.rq_encode(docs, cb, mode=mode)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        byts = 4.0 * n * bench.DIM + 4.0 * n * M
        res[mode] = {"ms": round(ms, 2), "rows_per_s": round(n / ms * 1e3), "algorithmic_gb_per_s": round(byts / ms / 1e6, 1),
                     "frac_of_hbm_peak": round(byts / ms / 1e6 / 8000.0, 4)}
        if mode == "fast":
            res[mode]["stats"] = {k: v for k, v in rq.last_encode_stats().items()}
            fast_codes = codes
        else:
            res["identical"] = bool(torch.equal(codes, fast_codes))
            res["rows_differing"] = int((codes != fast_codes).any(1).sum().item())
    out["rq_%dx%d" % (M, K)] = res
    print(json.dumps({("rq_%dx%d" % (M, K)): res}), flush=True)
    del codes, fast_codes, cb
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        json.dump(out, f, indent=1)
