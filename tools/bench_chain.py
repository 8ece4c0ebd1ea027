"""Config C4 of SURVEY.md 8(d) on one MI355X (tools/chain_c4.py; bench.py runs the same function after its timed region).
python tools/bench_chain.py [nq] [n_docs] [main.py --device_batch_size]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
import chain_c4  # noqa: E402
import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_QUERIES
N = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N_DOCS
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
M, K, R = 4, 32, 10
dev = torch.device("cuda:0")
model, tower, g, rn = synth.build(dev, M, K, batch)
tower.batch_size = None          # the product's default pass size (mevi_amd.t5.DEVICE_PASS_TOKENS)
rng = np.random.default_rng(0)
ids, mask = synth.query_ids(nq, dev, rng)
docs = bench.gen_shard(0, N, dev, N)
out, _ = chain_c4.run(model, tower, docs, ids, mask, bench.planted_ids(nq, N), rn, M, K, R, bench.TOPK, batch, rng, quiet=False)
for k, v in out["ms"].items():
    print(f"{k:24s} {v:9.1f} ms   {nq / v * 1e3:9.0f} q/s")
print(f"{'chain':24s} {out['chain_ms']:9.1f} ms   {out['queries_per_s']:9.0f} q/s   ({out['fine_candidates_per_query']:.1f} fine "
      f"candidates/query, max {out['fine_candidates_max']})")
print(json.dumps(out))
