#!/usr/bin/env python3
"""BERT-family tower at bert-base shapes (12 x 768, 12 heads, ff 3072): queries (32 tokens) and passages (128)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import bert, hip  # noqa: E402

if os.environ.get("MEVI_PROBE_LIB"):
    hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
d, ff, H, NL, V = 768, 3072, 12, 12, 30522


def rn(*shape, s=1.0):
    return torch.randn(shape, device=dev, generator=g) * s


W = {"embeddings.word_embeddings.weight": rn(V, d, s=0.02), "embeddings.position_embeddings.weight": rn(512, d, s=0.02),
     "embeddings.token_type_embeddings.weight": rn(2, d, s=0.02), "embeddings.LayerNorm.weight": torch.ones(d, device=dev),
     "embeddings.LayerNorm.bias": torch.zeros(d, device=dev)}
for l in range(NL):
    p = f"encoder.layer.{l}."
    for n_ in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
        W[p + n_ + ".weight"], W[p + n_ + ".bias"] = rn(d, d, s=d ** -0.5), rn(d, s=0.02)
    W[p + "intermediate.dense.weight"], W[p + "intermediate.dense.bias"] = rn(ff, d, s=d ** -0.5), rn(ff, s=0.02)
    W[p + "output.dense.weight"], W[p + "output.dense.bias"] = rn(d, ff, s=ff ** -0.5), rn(d, s=0.02)
    for n_ in ("attention.output.LayerNorm", "output.LayerNorm"):
        W[p + n_ + ".weight"], W[p + n_ + ".bias"] = torch.ones(d, device=dev), torch.zeros(d, device=dev)
tower = bert.BertTower(W, NL, H, device=dev, batch_size=512)
rng = np.random.default_rng(0)
for name, n, S, mean in (("queries", 2048, 32, 9), ("passages", 2048, 128, 70)):
    ids = np.zeros((n, S), np.int64)
    mask = np.zeros((n, S), np.int64)
    for i in range(n):
        L = int(np.clip(rng.normal(mean, mean / 2), 3, S))
        ids[i, :L] = rng.integers(5, V, size=L)
        mask[i, :L] = 1
    items = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev)}
    tower.encode_query(items)
    torch.cuda.synchronize()
    t = time.perf_counter()
    tower.encode_query(items)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    flop = NL * S * (2 * (4 * d * d + 2 * d * ff) + 4 * S * d)
    print(f"bert tower, {name}: {n} x {S} tokens in {dt*1e3:.1f} ms -> {n/dt:.0f} /s ({flop*n/dt/1e12:.1f} TFLOP/s padded-equivalent)", flush=True)
