#!/usr/bin/env python3
"""Passage side of the tower at t5-base shapes (gen_doc_embedding): passages/s on one GPU.
  python tools/bench_passage.py [n_passages] [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mevi_amd import hip, t5  # noqa: E402

if os.environ.get("MEVI_PROBE_LIB"):
    hip.LIB = os.path.abspath(os.environ["MEVI_PROBE_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
d, ff, H, dk, NL = 768, 3072, 12, 64, 12


def rn(*shape, s=1.0):
    return torch.randn(shape, device=dev, generator=g) * s


W = {"shared.weight": rn(32128, d)}
for stack, n_layers, dec in (("encoder", NL, False), ("decoder", NL, True)):
    for l in range(n_layers):
        p = f"{stack}.block.{l}.layer"
        for nm in "qkvo":
            W[f"{p}.0.SelfAttention.{nm}.weight"] = rn(d, d, s=d ** -0.5)
        W[f"{p}.0.layer_norm.weight"] = torch.ones(d, device=dev)
        i = 1
        if dec:
            for nm in "qkvo":
                W[f"{p}.1.EncDecAttention.{nm}.weight"] = rn(d, d, s=d ** -0.5)
            W[f"{p}.1.layer_norm.weight"] = torch.ones(d, device=dev)
            i = 2
        W[f"{p}.{i}.DenseReluDense.wi.weight"] = rn(ff, d, s=d ** -0.5)
        W[f"{p}.{i}.DenseReluDense.wo.weight"] = rn(d, ff, s=ff ** -0.5)
        W[f"{p}.{i}.layer_norm.weight"] = torch.ones(d, device=dev)
    W[f"{stack}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] = rn(32, H)
    W[f"{stack}.final_layer_norm.weight"] = torch.ones(d, device=dev)
tower = t5.TwinTower(W, device=dev, num_layers=NL, num_decoder_layers=NL, batch_size=batch)
rng = np.random.default_rng(0)
ids = np.zeros((n, 128), np.int64)
mask = np.zeros((n, 128), np.int64)
for i in range(n):
    L = int(np.clip(rng.normal(70, 30), 8, 128))
    ids[i, :L - 1] = rng.integers(3, 32100, size=L - 1)
    ids[i, L - 1] = 1
    mask[i, :L] = 1
psg = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev)}
tower.encode_passage(psg)
torch.cuda.synchronize()
t = time.perf_counter()
tower.encode_passage(psg)
torch.cuda.synchronize()
dt = time.perf_counter() - t
enc_flop = NL * 128 * (2 * (4 * d * d + 2 * d * ff) + 4 * 128 * d)
print(f"passage tower: {n} x 128 tokens (batch {batch}) in {dt*1e3:.1f} ms -> {n/dt:.0f} passages/s "
      f"({enc_flop*n/dt/1e12:.1f} TFLOP/s algorithmic, encoder)", flush=True)
