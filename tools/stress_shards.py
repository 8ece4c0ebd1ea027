#!/usr/bin/env python3
"""Randomised single-GPU simulation of the row-sharded search (mevi_amd.dense.sharded_ip_topk's rounds without the collective):
W shards searched in turn with the truncated first-round length, lists packed into the wire format, merged + proven by
mevi_topk_merge_packed_f32 (or the unfused merge where that does not fit), unproven queries repeated with full lists -- against the
un-sharded exact search, bit for bit.  Random W, shapes, k and row orders (incl. sorted: every top-k in one shard):
  python tools/stress_shards.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import dense  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
t0, cases, second, fused = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    W = int(rng.integers(2, 9))
    dim = int(rng.choice([64, 128, 768]))
    nd = int(rng.choice([50, 3000, 60_000, 500_000]) * rng.uniform(0.6, 1.4))
    nq = int(rng.choice([1, 5, 40, 300]))
    k = int(rng.choice([1, 10, 100, 1000]))
    order = str(rng.choice(["random", "sorted", "duplicates"]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    docs = torch.randn((nd, dim), device=dev, generator=g)
    q = torch.randn((nq, dim), device=dev, generator=g)
    if order == "sorted":
        docs = docs[torch.argsort(docs @ q[0], descending=True)].contiguous()
    elif order == "duplicates":
        docs = docs[:max(1, nd // 50)][torch.randint(0, max(1, nd // 50), (nd,), device=dev, generator=g)].contiguous()
    es, ei = dense.ip_topk(q, docs, k)
    shards = [dense.shard_range(nd, r, W) for r in range(W)]
    idx = [dense.DenseIndex(docs[a:b]) if b > a else None for a, b in shards]

    def round_(qq, kk):
        lists = []
        for (a, b), ix in zip(shards, idx):
            if ix is None:
                s = torch.full((qq.shape[0], kk), -torch.finfo(torch.float32).max, device=dev)
                i = torch.full((qq.shape[0], kk), -1, dtype=torch.int64, device=dev)
            else:
                s, i = ix.search(qq, kk, id_offset=a)
            lists.append(dense.pack_lists(s, i))
        gathered = torch.stack(lists)
        if dense._packed_merge_fits(W, kk, k):
            return dense.merge_packed(gathered, k), True
        all_s, all_i = dense.unpack_lists(gathered)
        return dense.merge_truncated(all_s, all_i, k), False

    kl = dense.truncated_list_len(k, W)
    (ms, mi, unproven), f = round_(q, kl)
    fused += int(f)
    redo = torch.nonzero(unproven).flatten()
    if redo.numel():
        second += 1
        (rs, ri, _), _ = round_(q[redo].contiguous(), k)
        ms[redo], mi[redo] = rs, ri
    ok = bool(torch.equal(mi, ei) and torch.equal(ms.view(torch.int32), es.view(torch.int32)))
    cases += 1
    if not ok:
        print("BAD", dict(W=W, dim=dim, nd=nd, nq=nq, k=k, order=order, kl=kl))
        sys.exit(1)
    del docs, q, idx
print(f"{cases} random sharded searches (W 2..8): merged lists identical to the un-sharded exact search; {second} needed the second round, "
      f"{fused} took the fused packed merge")
