#!/usr/bin/env python3
"""Randomised cross-check of the BERT-family tower (mevi_amd/bert.py, mtype 'bert': coCondenser / AR2 / ERNIE, MEVI/document_encoder.py:43-44,
104-123) against oracle/bert.py (pinned to the vendored BertModel by golden G8): the golden's architecture with RANDOM weights (its own
shapes and key names), 1..all layers, random batches of ragged lengths 1..max -- reps within 5e-5, packed == padded on real tokens:
  python tools/stress_bert.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mevi_amd import bert, nci  # noqa: E402
from oracle import bert as obert  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_bert_tower.npz"))
cfg0 = json.loads(str(g["cfg"]))
W0 = nci.load_npz_weights(g)
S, vocab = g["input_ids"].shape[1], int(W0[[k for k in W0 if k.endswith("word_embeddings.weight")][0]].shape[0])
t0, cases, worst = time.time(), 0, 0.0
while time.time() - t0 < budget:
    torch.manual_seed(int(rng.integers(1 << 30)))
    L = int(rng.integers(1, cfg0["num_hidden_layers"] + 1))
    W = {}
    for k, v in W0.items():
        if "LayerNorm.weight" in k:
            W[k] = 1 + 0.1 * torch.randn_like(v)
        elif "LayerNorm.bias" in k or k.endswith(".bias"):
            W[k] = 0.05 * torch.randn_like(v)
        else:
            W[k] = torch.randn_like(v) * (float(v.std()) if v.numel() > 1 else 1.0) * float(rng.choice([0.5, 1.0, 2.0]))
    cfg = dict(cfg0, num_hidden_layers=L)
    B = int(rng.integers(1, 20))
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), np.int64)
    for i in range(B):
        n = int(rng.choice([1, 2, S, int(rng.integers(1, S + 1))]))
        ids[i, :n] = rng.integers(1, vocab, size=n)
        mask[i, :n] = 1
    ids, mask = torch.from_numpy(ids), torch.from_numpy(mask)
    tower = bert.BertTower(W, L, cfg["num_attention_heads"], eps=cfg["layer_norm_eps"], device=dev)
    reps = tower.encode_query({"input_ids": ids, "attention_mask": mask}).cpu()
    with torch.no_grad():
        want = obert.tower_encode({k: v for k, v in W.items()}, cfg, ids, mask)
    diff = float((reps - want).abs().max() / max(1.0, float(want.abs().max())))
    worst = max(worst, diff)
    hid = tower.lm_q.forward(ids.to(dev), mask.to(dev), pack=False).cpu().numpy()
    packed = tower.lm_q.forward(ids.to(dev), mask.to(dev), pack=True).cpu().numpy()
    valid = mask.numpy().astype(bool)
    if diff > 5e-5 or not np.array_equal(packed[valid], hid[valid]):
        print("BAD", dict(L=L, B=B, diff=diff))
        sys.exit(1)
    cases += 1
    del tower
print(f"{cases} random BERT towers: reps within {worst:.2e} (relative) of the oracle, packed == padded on the real tokens")
